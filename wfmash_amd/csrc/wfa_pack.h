// wfa_pack.h -- the 2-bit packed form of the sequences (wfa_tile2.hip) in one place, for the device and for the host:
// the code of a base, the word layout of the mirror, 16 / 32 bases from an arbitrary base offset, the first difference of
// two packed words.  The kernels use these primitives; the host model below them (pk_lce_model: the staged extension, 16 bases,
// then 64, then 32 at a time) is what the CPU suite holds against a byte-wise comparison (tests/test_pack_cpu.py).
#ifndef WFM_WFA_PACK_H_
#define WFM_WFA_PACK_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define WFM_HD __host__ __device__ __forceinline__
#else
#define WFM_HD inline
#endif

namespace wfm {

// base at byte index a of the sequence buffer = bits 2 (a & 15) .. of word a >> 4; code (c >> 1) & 3: A 0, C 1, T 2, G 3
WFM_HD uint32_t pack_code(uint8_t c) { return ((uint32_t)c >> 1) & 3u; }
WFM_HD bool pack_is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// (hi:lo) >> (sh & 31), low 32 bits -- v_alignbit_b32
WFM_HD uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  sh &= 31u;
  return sh ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : lo;
#endif
}

// 16 / 32 bases from base offset o of a packed word array (any pointer type: LDS window, global mirror, host array)
template <typename W>
WFM_HD uint32_t pk16(W w, uint32_t o) {
  const W q = w + (o >> 4);
  return alignbit32(q[1], q[0], o << 1);
}
template <typename W>
WFM_HD uint64_t pk32(W w, uint32_t o) {
  const W q = w + (o >> 4);
  const uint32_t a = q[0], b = q[1], c = q[2];
  const uint32_t sh = o << 1;
  return ((uint64_t)alignbit32(c, b, sh) << 32) | alignbit32(b, a, sh);
}
// number of leading bases (of 16) on which two packed words agree: x = their xor
WFM_HD uint32_t first_diff16(uint32_t x) { return (uint32_t)(x ? __builtin_ctz(x) : 32) >> 1; }

#if defined(__HIPCC__)
#define WFM_HOST_ONLY __host__ inline
#else
#define WFM_HOST_ONLY inline
#endif
// ---- host model (CPU test-suite) ----
// the mirror of n bytes: (n + 15) / 16 words (+ the caller's padding)
WFM_HOST_ONLY void pack_words_model(const uint8_t* seq, int64_t n, uint32_t* out) {
  const int64_t nw = (n + 15) / 16;
  for (int64_t i = 0; i < nw; ++i) {
    uint32_t v = 0;
    for (int j = 0; j < 16; ++j) {
      const int64_t a = i * 16 + j;
      v |= pack_code(a < n ? seq[a] : (uint8_t)0) << (2 * j);
    }
    out[i] = v;
  }
}
// the staged extension of one cell: bases that agree from offsets (oP, oT) of two packed arrays on, at most maxn
WFM_HOST_ONLY int pk_lce_model(const uint32_t* wP, const uint32_t* wT, uint32_t oP, uint32_t oT, int maxn) {
  if (maxn <= 0) return 0;
  const uint32_t n16 = first_diff16(pk16(wP, oP) ^ pk16(wT, oT));
  if (n16 < 16u || maxn <= 16) return (int)n16 < maxn ? (int)n16 : maxn;
  int n = 80;
  for (int q = 0; q < 4; ++q) {  // bases 16 .. 79: pk_stage2
    const uint32_t x = pk16(wP, oP + 16 + 16 * (uint32_t)q) ^ pk16(wT, oT + 16 + 16 * (uint32_t)q);
    if (x) { n = 16 + 16 * q + (int)(__builtin_ctz(x) >> 1); break; }
  }
  if (n < 80 || maxn <= 80) return n < maxn ? n : maxn;
  for (int at = 80;; at += 32) {  // the wave's tail, one lane's share at a time
    if (at >= maxn) return maxn;
    const uint64_t x = pk32(wP, oP + (uint32_t)at) ^ pk32(wT, oT + (uint32_t)at);
    if (x) { const int r = at + (int)(__builtin_ctzll(x) >> 1); return r < maxn ? r : maxn; }
  }
}

}  // namespace wfm
#endif
