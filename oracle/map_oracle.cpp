// oracle/map_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
// CPU restatement of the mashmap3 sketching primitives of the map path:
//   m1  CommonFunc::getHash            src/map/include/commonFunc.hpp:173-182
//       MurmurHash3_x64_128 (seed 42)  src/common/murmur3.h:226-302
//   m2  makeUpperCaseAndValidDNA       commonFunc.hpp:132-142 ; reverseComplement :74-83
//   m5  CommonFunc::sketchSequence     commonFunc.hpp:218-323
// Pinned against the reference itself: oracle/_ref/libref_map.so is the reference's own
// commonFunc.hpp compiled in place (oracle/ref_map.cpp), and tests/golden/map_*.json hold
// vectors generated from it (tests/golden/make_map_golden.py).
// Only tests/, smoke() and bench.py's cpu_baseline leg may use this file.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <unordered_map>
#include <vector>

namespace {

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

// murmur3.h:226-302, low 64 bits of the 128-bit digest
uint64_t murmur3_x64_128_lo(const uint8_t* data, int len, uint32_t seed) {
  const int nblocks = len / 16;
  uint64_t h1 = seed, h2 = seed;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  for (int i = 0; i < nblocks; ++i) {
    uint64_t k1, k2;
    memcpy(&k1, data + 16 * i, 8);
    memcpy(&k2, data + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t* tail = data + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  const int t = len & 15;
  for (int i = t - 1; i >= 8; --i) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
  if (t > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (int i = std::min(t, 8) - 1; i >= 0; --i) k1 ^= (uint64_t)tail[i] << (8 * i);
  if (t > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

inline char norm_base(char c) {
  if (c > 96 && c < 123) c -= 32;
  return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';
}
inline char comp_base(char c) {
  switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }
}

}  // namespace

extern "C" {

struct mo_minmer_t { uint64_t hash; int64_t wpos, wpos_end; int32_t seqId; int16_t strand; int16_t pad; };

uint64_t mo_get_hash(const char* seq, int len) { return murmur3_x64_128_lo((const uint8_t*)seq, len, 42u); }

void mo_upper_valid(char* seq, int64_t len) { for (int64_t i = 0; i < len; ++i) seq[i] = norm_base(seq[i]); }

void mo_revcomp(const char* src, char* dst, int len) { for (int i = 0; i < len; ++i) dst[len - 1 - i] = comp_base(src[i]); }

// canonical hash + strand of every k-mer start (the contract of wfm_hash_kmers)
void mo_hash_kmers(const char* seq_in, int64_t len, int k, uint64_t* hash, int8_t* strand) {
  std::vector<char> seq(seq_in, seq_in + len), rc(k);
  mo_upper_valid(seq.data(), len);
  for (int64_t i = 0; i + k <= len; ++i) {
    bool has_n = false;
    for (int j = 0; j < k; ++j) has_n |= seq[i + j] == 'N';
    uint64_t h = ~0ull; int8_t st = 0;
    if (!has_n) {
      mo_revcomp(seq.data() + i, rc.data(), k);
      const uint64_t hf = mo_get_hash(seq.data() + i, k), hb = mo_get_hash(rc.data(), k);
      if (hf != hb) { h = std::min(hf, hb); st = hf < hb ? 1 : -1; }
    }
    hash[i] = h; strand[i] = st;
  }
}

// sketchSequence restated with the reference's streaming structure: a max-heap of the current
// <= s smallest distinct canonical hashes plus a hash -> MinmerInfo table (commonFunc.hpp:238-321).
int mo_sketch_sequence(const char* seq_in, int64_t len, int k, int s, int32_t seq_id, mo_minmer_t* out, int cap) {
  std::vector<char> seq(seq_in, seq_in + len), rev(len);
  mo_upper_valid(seq.data(), len);
  mo_revcomp(seq.data(), rev.data(), (int)len);
  std::unordered_map<uint64_t, mo_minmer_t> vals;
  std::vector<uint64_t> heap;  // max-heap
  int ambig = 0;
  for (int i = k - 1; i >= 0; --i) if (i < len && seq[i] == 'N') { ambig = i + 1; break; }
  for (int64_t i = 0; i + k <= len; ++i) {
    if (seq[i + k - 1] == 'N') ambig = k;
    const uint64_t hf = mo_get_hash(seq.data() + i, k);
    const uint64_t hb = mo_get_hash(rev.data() + len - i - k, k);
    if (hb != hf && ambig == 0) {
      const uint64_t cur = std::min(hf, hb);
      const int16_t cs = hf < hb ? 1 : -1;
      if ((int)heap.size() < s || cur <= heap.front()) {
        if (heap.empty() || vals.find(cur) == vals.end()) {
          if ((int)vals.size() < s || cur < heap.front()) {
            vals[cur] = mo_minmer_t{cur, i, i, seq_id, cs, 0};
            heap.push_back(cur);
            std::push_heap(heap.begin(), heap.end());
          }
          if ((int)vals.size() > s) {
            vals.erase(heap.front());
            std::pop_heap(heap.begin(), heap.end());
            heap.pop_back();
          }
        } else {
          vals[cur].wpos_end = i;
          vals[cur].strand += cs;
        }
      }
    }
    if (ambig > 0) --ambig;
  }
  const int n = (int)heap.size();
  std::vector<mo_minmer_t> res(n);
  for (int r = n - 1; r >= 0; --r) {
    mo_minmer_t m = vals[heap.front()];
    m.strand = m.strand > 0 ? 1 : (m.strand == 0 ? 0 : -1);
    res[r] = m;
    std::pop_heap(heap.begin(), heap.end());
    heap.pop_back();
  }
  for (int r = 0; r < n && r < cap; ++r) out[r] = res[r];
  return n;
}

}  // extern "C"
