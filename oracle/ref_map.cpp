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
// streamingMinHash.hpp also compiles unmodified: StreamingMinHash is the sketch Stat::estimate_identity_for_groups
// keeps per worker and per group (map_stats.hpp:540-660), and GroupedStreamingMinHash::processSequence is the same
// k-mer loop (ambiguity counter, canonical hash, palindromes dropped) as map_stats.hpp:569-616, which itself cannot be
// compiled here (GSL, htslib).
#include "map/include/streamingMinHash.hpp"

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

// stderr parked on /dev/null for a whole multi-threaded leg (bench.py's map baseline runs 64 calls side by side: parking it per call let one
// call's restore show the others' meters -- "ref [0.3% complete ..." all over the driver's stderr tail).  on = 1 parks, 0 restores.
static int g_parked_fd = -1;
void ref_park_stderr(int on) {
  fflush(stderr);
  if (on && g_parked_fd < 0) {
    g_parked_fd = dup(2);
    const int nul = open("/dev/null", O_WRONLY);
    if (nul >= 0) { dup2(nul, 2); close(nul); }
  } else if (!on && g_parked_fd >= 0) {
    dup2(g_parked_fd, 2); close(g_parked_fd); g_parked_fd = -1;
  }
}

// CommonFunc::addMinmers (commonFunc.hpp:440-708).  Returns the number of minmer intervals.
int64_t ref_add_minmers(char* seq, int64_t len, int k, int w, int s, int32_t seq_id, ref_minmer_t* out, int64_t cap) {
  std::vector<skch::MinmerInfo> v;
  {
    // addMinmers calls progress->increment unconditionally (commonFunc.hpp:479); the meter prints
    // to stderr from its own thread, so stderr is parked on /dev/null for the duration of the call
    // (unless the caller has parked it for a whole leg: ref_park_stderr)
    const bool own = g_parked_fd < 0;
    int saved = -1, nul = -1;
    if (own) { fflush(stderr); saved = dup(2); nul = open("/dev/null", O_WRONLY); dup2(nul, 2); }
    {
      progress_meter::ProgressMeter pm((uint64_t)len * 100 + 100, "ref", false);
      skch::CommonFunc::addMinmers(v, seq, len, k, w, 4, s, seq_id, &pm);
      pm.finish();
    }
    if (own) { fflush(stderr); dup2(saved, 2); close(saved); close(nul); }
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

// Bottom-`sketch_size` MinHash of a set of sequences pooled by group, by the reference's own classes:
// GroupedStreamingMinHash::processSequence per sequence (streamingMinHash.hpp:167-241), sketches out in ascending order
// (StreamingMinHash::getSketch).  groups[i] = group of sequence i; out receives the sketch of `want_group`.
int64_t ref_group_minhash(const char* const* seqs, const int64_t* lens, const int32_t* groups, int n, int k, int sketch_size,
                          int32_t want_group, uint64_t* out, int64_t cap) {
  skch::GroupedStreamingMinHash g((size_t)sketch_size, 0);
  for (int i = 0; i < n; ++i)
    if (lens[i] >= k) g.processSequence(seqs[i], lens[i], i, groups[i], k, 4, nullptr);
  const auto all = g.getAllGroupSketches();
  const auto it = all.find(want_group);
  if (it == all.end()) return 0;
  int64_t m = 0;
  for (uint64_t h : it->second) { if (m < cap) out[m] = h; ++m; }
  return m;
}

// StreamingMinHash alone: add the values in order, return the sketch (keeps duplicates; a value equal to the
// current maximum does not enter a full sketch)
int64_t ref_streaming_minhash(const uint64_t* values, int64_t n, int sketch_size, uint64_t* out, int64_t cap) {
  skch::StreamingMinHash mh((size_t)sketch_size, 0);
  for (int64_t i = 0; i < n; ++i) mh.add_unsafe(values[i]);
  const auto v = mh.getSketch();
  int64_t m = 0;
  for (uint64_t h : v) { if (m < cap) out[m] = h; ++m; }
  return m;
}

}  // extern "C"
