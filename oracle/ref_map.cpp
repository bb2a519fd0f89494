// oracle/ref_map.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin C wrappers around the REFERENCE'S OWN map-path header, compiled where it lies:
//   g++ -I/root/reference/src -I/root/reference/src/common  (oracle/Makefile, target `ref`)
// Nothing of the reference is copied into this repository; the resulting
// oracle/_ref/libref_map.so is git-ignored and only built where /root/reference exists.
// commonFunc.hpp (+ murmur3.h, base_types.hpp, map_parameters.hpp, ankerl, progress.hpp)
// compiles unmodified and needs no stand-in header.
#include <fcntl.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "map/include/base_types.hpp"
#include "map/include/commonFunc.hpp"

extern "C" {

// CommonFunc::getHash (commonFunc.hpp:173-182)
uint64_t ref_get_hash(const char* seq, int len) { return skch::CommonFunc::getHash(seq, len); }

// makeUpperCaseAndValidDNA (commonFunc.hpp:132-142), in place
void ref_upper_valid(char* seq, int64_t len) { skch::CommonFunc::makeUpperCaseAndValidDNA(seq, len); }

// reverseComplement (commonFunc.hpp:74-83)
void ref_revcomp(const char* src, char* dst, int len) { skch::CommonFunc::reverseComplement(src, dst, len); }

struct ref_minmer_t { uint64_t hash; int64_t wpos, wpos_end; int32_t seqId; int16_t strand; int16_t pad; };

// CommonFunc::sketchSequence (commonFunc.hpp:218-323).  seq is modified in place (upper-cased).
int ref_sketch_sequence(char* seq, int64_t len, int k, int s, int32_t seq_id, ref_minmer_t* out, int cap) {
  std::vector<skch::MinmerInfo> v;
  skch::CommonFunc::sketchSequence(v, seq, len, k, 4, s, seq_id);
  int n = 0;
  for (const auto& m : v) {
    if (n >= cap) break;
    out[n].hash = m.hash; out[n].wpos = m.wpos; out[n].wpos_end = m.wpos_end; out[n].seqId = m.seqId;
    out[n].strand = m.strand; out[n].pad = 0;
    ++n;
  }
  return (int)v.size();
}

// CommonFunc::addMinmers (commonFunc.hpp:440-708).  Returns the number of minmer intervals.
int64_t ref_add_minmers(char* seq, int64_t len, int k, int w, int s, int32_t seq_id, ref_minmer_t* out, int64_t cap) {
  std::vector<skch::MinmerInfo> v;
  {
    // addMinmers calls progress->increment unconditionally (commonFunc.hpp:479); the meter prints
    // to stderr from its own thread, so stderr is parked on /dev/null for the duration of the call
    fflush(stderr);
    const int saved = dup(2), nul = open("/dev/null", O_WRONLY);
    dup2(nul, 2);
    {
      progress_meter::ProgressMeter pm((uint64_t)len * 100 + 100, "ref", false);
      skch::CommonFunc::addMinmers(v, seq, len, k, w, 4, s, seq_id, &pm);
      pm.finish();
    }
    fflush(stderr);
    dup2(saved, 2); close(saved); close(nul);
  }
  int64_t n = 0;
  for (const auto& m : v) {
    if (n >= cap) break;
    out[n].hash = m.hash; out[n].wpos = m.wpos; out[n].wpos_end = m.wpos_end; out[n].seqId = m.seqId;
    out[n].strand = m.strand; out[n].pad = 0;
    ++n;
  }
  return (int64_t)v.size();
}

}  // extern "C"
