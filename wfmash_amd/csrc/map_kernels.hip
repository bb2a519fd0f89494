// map_kernels.hip -- mashmap3 sketching kernels for gfx950 (map path, SURVEY 8a m1,m2,m5).
//
//   normalize_kernel        makeUpperCaseAndValidDNA   (commonFunc.hpp:132-142)
//   kmer_hash_kernel        getHash fwd + revcomp, canonical min, strand
//                           (commonFunc.hpp:173-182, murmur3.h:226-302, seed 42)
//   sketch_fragments_kernel sketchSequence: bottom-s distinct canonical hashes of
//                           one fragment per workgroup (commonFunc.hpp:218-323)
//
// HBM-bound byte/integer work: 1 B/base read; k-mer words come from L1/L2 via
// unaligned 8-byte loads of the normalised buffer; per-fragment selection is a
// bitonic sort of (hash,pos) in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>  // rocprim's texture iterator calls host memset
#include <algorithm>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "wfa_handle.h"
#include "dev_cache.h"
#include "map_device.h"

namespace wfm {

__device__ __forceinline__ uint64_t ld8(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

// MurmurHash3_x64_128 (low 64 bits) of a key of len <= 32 bytes held
// little-endian in w[0..3] with bytes >= len zeroed.  murmur3.h:226-302.
__device__ __forceinline__ uint64_t murmur3_x64_lo(const uint64_t w[4], int len, uint32_t seed) {
  uint64_t h1 = seed, h2 = seed;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  const int nblocks = len >> 4;
  for (int i = 0; i < nblocks; ++i) {
    uint64_t k1 = w[2 * i], k2 = w[2 * i + 1];
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const int tail = len & 15;
  if (tail) {
    uint64_t k1 = nblocks == 0 ? w[0] : w[2], k2 = nblocks == 0 ? w[1] : w[3];
    if (tail > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

__device__ __forceinline__ uint8_t norm_base(uint8_t c) {
  if (c > 96 && c < 123) c -= 32;
  return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : (uint8_t)'N';
}

// byte j of a k-mer held in w[]
__device__ __forceinline__ uint32_t kbyte(const uint64_t w[4], int j) { return (uint32_t)(w[j >> 3] >> ((j & 7) * 8)) & 0xffu; }

// Loads the k-mer at p (k <= 32) into fw[], builds its reverse complement in
// rc[] (commonFunc.hpp:74-83).  Returns false if it contains an N.
__device__ __forceinline__ bool load_kmer(const uint8_t* p, int k, uint64_t fw[4], uint64_t rc[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rem = k - 8 * q;
    uint64_t v = rem > 0 ? ld8(p + 8 * q) : 0ull;
    if (rem > 0 && rem < 8) v &= (~0ull) >> (64 - 8 * rem);
    fw[q] = v;
    rc[q] = 0;
  }
  bool ok = true;
  const uint32_t lut = (uint32_t)'A' | ((uint32_t)'C' << 8) | ((uint32_t)'T' << 16) | ((uint32_t)'G' << 24);
  for (int j = 0; j < k; ++j) {
    const uint32_t b = kbyte(fw, j);
    ok = ok && (b != 'N');
    const uint32_t code = ((b >> 1) & 3u) ^ 2u;  // A0 C1 T2 G3 ; complement = ^2
    const uint64_t cb = (lut >> (8 * code)) & 0xffu;
    const int d = k - 1 - j;
    rc[d >> 3] |= cb << ((d & 7) * 8);
  }
  return ok;
}

__global__ void normalize_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i0 + 16 <= n) {
    uint4 v = *reinterpret_cast<const uint4*>(in + i0);
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t x = w[q], y = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) y |= (uint32_t)norm_base((uint8_t)(x >> (8 * b))) << (8 * b);
      w[q] = y;
    }
    *reinterpret_cast<uint4*>(out + i0) = v;
  } else {
    for (int64_t i = i0; i < n; ++i) out[i] = norm_base(in[i]);
  }
}

__global__ void kmer_hash_kernel(const uint8_t* __restrict__ seq /*normalised, padded*/, int64_t nk, int k,
                                 uint64_t* __restrict__ hash, int8_t* __restrict__ strand) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t fw[4], rc[4];
    const bool ok = load_kmer(seq + i, k, fw, rc);
    uint64_t h = ~0ull;
    int8_t st = 0;
    if (ok) {
      const uint64_t hf = murmur3_x64_lo(fw, k, 42u), hb = murmur3_x64_lo(rc, k, 42u);
      if (hf != hb) { h = hf < hb ? hf : hb; st = hf < hb ? 1 : -1; }
    }
    hash[i] = h;
    strand[i] = st;
  }
}

// The same for 9 <= K <= 24, the k-mer in two or three words and nothing byte by byte: the reverse complement is the
// complement of every byte -- A <-> T differ in bits 0, 2, 4, C <-> G in bit 2, and bit 1 tells the two pairs apart --
// followed by a reversal of the bytes and a shift; an N is a zero byte of word ^ 'NNNNNNNN'.
template <int K>
__device__ __forceinline__ void kmer_hash_2w(const uint8_t* __restrict__ p, uint64_t& h, int8_t& st) {
  static_assert(K >= 9 && K <= 24, "two or three words");
  constexpr int NWD = (K + 7) / 8;                       // words that hold the k-mer
  constexpr int LASTB = K - 8 * (NWD - 1);               // bytes of the last one
  constexpr uint64_t last_mask = LASTB == 8 ? ~0ull : ((~0ull) >> (64 - 8 * LASTB));
  constexpr uint64_t ones = 0x0101010101010101ULL;
  constexpr int sh = 8 * (8 * NWD - K);                  // bits the reversed words are shifted down by
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t f[3] = {0, 0, 0};
  uint64_t nbits = 0;
#pragma unroll
  for (int q = 0; q < NWD; ++q) {
    f[q] = ld8(p + 8 * q);
    if (q == NWD - 1) f[q] &= last_mask;
    const uint64_t x = f[q] ^ (ones * 'N');
    nbits |= (x - ones) & ~x;
  }
  h = ~0ull;
  st = 0;
  if (!(nbits & (ones * 0x80))) {
    // complement, reverse the bytes, drop the bytes that were past the k-mer
    uint64_t b[3] = {0, 0, 0}, r[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < NWD; ++q) {
      const uint64_t m = (~f[q] >> 1) & ones;
      b[NWD - 1 - q] = __builtin_bswap64(f[q] ^ (ones * 4) ^ (m * 0x11));
    }
#pragma unroll
    for (int q = 0; q < NWD; ++q) r[q] = sh == 0 ? b[q] : ((b[q] >> sh) | (q + 1 < NWD ? b[q + 1] << ((64 - sh) & 63) : 0ull));
    uint64_t hv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const uint64_t* w = d == 0 ? f : r;
      uint64_t h1 = 42u, h2 = 42u;
      if (K >= 16) {  // one whole block
        uint64_t k1 = w[0], k2 = w[1];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
      }
      constexpr int tail = K & 15;
      if (tail) {
        uint64_t k1 = K >= 16 ? w[2] : w[0], k2 = K >= 16 ? 0ull : w[1];
        if (tail > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
      }
      h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
      h1 += h2; h2 += h1;
      h1 = fmix64(h1); h2 = fmix64(h2);
      hv[d] = h1 + h2;
    }
    if (hv[0] != hv[1]) { h = hv[0] < hv[1] ? hv[0] : hv[1]; st = hv[0] < hv[1] ? 1 : -1; }
  }
}
template <int K>
__global__ void __launch_bounds__(256) kmer_hash_2w_kernel(const uint8_t* __restrict__ seq /*normalised, padded*/, int64_t nk, uint64_t* __restrict__ hash,
                                                           int8_t* __restrict__ strand) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t h; int8_t st;
    kmer_hash_2w<K>(seq + i, h, st);
    hash[i] = h;
    strand[i] = st;
  }
}

// canonical hash + strand of every k-mer start of a normalised, padded sequence
// Hashing and thresholding in one pass (the MinHash sketch of a chromosome, wfm_minhash_sketch): the canonical hashes at most tau -- a few
// ten thousand of 2.5 * 10^8 -- are appended to `out`, nothing else is written.  The hash array of the two-pass form (2 GB written, then read
// again by the select) never exists.  k-mers before `skip_first` are left out (an ambiguous base among the first k bases blanks them).
template <int K>
__global__ void __launch_bounds__(256) kmer_hash_select_kernel(const uint8_t* __restrict__ seq /*normalised, padded*/, int64_t nk, int64_t skip_first, uint64_t tau,
                                                               uint64_t* __restrict__ out, unsigned long long* __restrict__ count, unsigned long long cap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t h; int8_t st;
    kmer_hash_2w<K>(seq + i, h, st);
    if (i >= skip_first && h <= tau) {
      const unsigned long long at = atomicAdd(count, 1ull);
      if (at < cap) out[at] = h;
    }
  }
}
// false: no fused form for this k (the caller takes the two-pass path)
static bool launch_kmer_hash_select(const uint8_t* d_norm, int64_t nk, int k, int64_t skip_first, uint64_t tau, uint64_t* d_out, unsigned long long* d_count,
                                    unsigned long long cap, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((nk + 255) / 256, 256 * 64);
  switch (k) {
#define WFM_K2S(K) case K: hipLaunchKernelGGL(kmer_hash_select_kernel<K>, dim3(blocks), dim3(256), 0, st, d_norm, nk, skip_first, tau, d_out, d_count, cap); return true;
    WFM_K2S(15) WFM_K2S(16) WFM_K2S(17) WFM_K2S(19) WFM_K2S(21)
#undef WFM_K2S
    default: return false;
  }
}

static void launch_kmer_hash(const uint8_t* d_norm, int64_t nk, int k, uint64_t* d_hash, int8_t* d_strand, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((nk + 255) / 256, 256 * 8);
  switch (k) {
#define WFM_K2W(K) case K: hipLaunchKernelGGL(kmer_hash_2w_kernel<K>, dim3(blocks), dim3(256), 0, st, d_norm, nk, d_hash, d_strand); break;
    WFM_K2W(9) WFM_K2W(10) WFM_K2W(11) WFM_K2W(12) WFM_K2W(13) WFM_K2W(14) WFM_K2W(15) WFM_K2W(16)
    WFM_K2W(17) WFM_K2W(18) WFM_K2W(19) WFM_K2W(20) WFM_K2W(21) WFM_K2W(22) WFM_K2W(23) WFM_K2W(24)
#undef WFM_K2W
    default: hipLaunchKernelGGL(kmer_hash_kernel, dim3(blocks), dim3(256), 0, st, d_norm, nk, k, d_hash, d_strand);
  }
}

// One workgroup per fragment.  LDS: key[npow2] (u64) + pv[npow2] (u32: pos<<1 | isRev).
__global__ __launch_bounds__(256) void sketch_fragments_kernel(const uint8_t* __restrict__ seq, const int64_t* __restrict__ frag_off,
                                                               const int32_t* __restrict__ frag_len, int k, int s, int32_t seq_id,
                                                               int npow2, wfm_minmer_t* __restrict__ out, int32_t* __restrict__ out_count,
                                                               uint64_t* gkey, uint32_t* gpv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // the (hash, position) pairs of the fragment: in LDS, or -- fragments of more k-mers than LDS holds (windows above 8 k) --
  // in a slice of global memory; the sort and the selection below are the same either way
  uint64_t* key = gkey ? gkey + (size_t)f * npow2 : reinterpret_cast<uint64_t*>(smem);
  uint32_t* pv = gpv ? gpv + (size_t)f * npow2 : reinterpret_cast<uint32_t*>(smem + (size_t)npow2 * 8);
  int* ssum = reinterpret_cast<int*>(smem + (gkey ? 0 : (size_t)npow2 * 12));  // [s] strand sums
  int* sbase = ssum + s;                                                       // [4] wave scan totals + running base
  const uint8_t* p = seq + frag_off[f];
  const int len = frag_len[f];
  const int nk = len - k + 1;
  for (int i = tid; i < npow2; i += blockDim.x) {
    uint64_t h = ~0ull; uint32_t v = 0xffffffffu;
    if (i < nk) {
      uint64_t fw[4], rc[4];
      if (load_kmer(p + i, k, fw, rc)) {
        const uint64_t hf = murmur3_x64_lo(fw, k, 42u), hb = murmur3_x64_lo(rc, k, 42u);
        if (hf != hb) { h = hf < hb ? hf : hb; v = ((uint32_t)i << 1) | (hf < hb ? 0u : 1u); }
      }
    }
    key[i] = h; pv[i] = v;
  }
  for (int i = tid; i < s; i += blockDim.x) ssum[i] = 0;
  if (tid < 8) sbase[tid] = 0;
  __syncthreads();
  // bitonic sort ascending by (key, pv)
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (npow2 >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool asc = ((lo & size) == 0);
        const uint64_t ka = key[lo], kb = key[hi];
        const uint32_t va = pv[lo], vb = pv[hi];
        const bool gt = (ka > kb) || (ka == kb && va > vb);
        if (gt == asc) { key[lo] = kb; key[hi] = ka; pv[lo] = vb; pv[hi] = va; }
      }
      __syncthreads();
    }
  }
  // distinct rank of every run head (ordered block scan over chunks of blockDim)
  int running = 0;
  for (int i0 = 0; i0 < npow2; i0 += blockDim.x) {
    const int i = i0 + tid;
    const uint64_t ki = key[i];
    const bool valid = ki != ~0ull || pv[i] != 0xffffffffu;
    const bool head = valid && (i == 0 || key[i - 1] != ki);
    const unsigned long long m = __ballot(head);
    const int pre = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sbase[wid] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w2 = 0; w2 < 4; ++w2) { const int x = sbase[w2]; if (w2 < wid) woff += x; tot += x; }
    // rank of the run this element belongs to = (#heads at or before i) - 1
    const int heads_incl = running + woff + pre + (head ? 1 : 0);
    const int r = heads_incl - 1;
    if (valid && r < s) {
      const bool tail = (i + 1 >= npow2) || key[i + 1] != ki;
      wfm_minmer_t* o = out + (size_t)f * s + r;
      if (head) { o->hash = ki; o->wpos = (int64_t)(pv[i] >> 1); o->seqId = seq_id; o->pad_ = 0; }
      if (tail) o->wpos_end = (int64_t)(pv[i] >> 1);
      atomicAdd(&ssum[r], (pv[i] & 1u) ? -1 : 1);
    }
    running += tot;
    __syncthreads();
  }
  const int cnt = min(running, s);
  for (int r = tid; r < cnt; r += blockDim.x) {
    const int v = ssum[r];
    out[(size_t)f * s + r].strand = (int16_t)(v > 0 ? 1 : (v == 0 ? 0 : -1));  // FWD=1, AMBIG=0, REV=-1
  }
  if (tid == 0) out_count[f] = cnt;
}

// sketchSequence by threshold and table (round 3).  The sort above orders all ~5000 (hash, position) pairs of a fragment to
// find its ~25 smallest distinct hashes, in 60 - 100 KB of LDS (one workgroup per CU).  Here a threshold tau is set so that
// about 2 s + 24 distinct hashes are expected at or below it (a canonical hash is the smaller of two uniform values: a
// fraction t of the range holds ~2 t of the k-mers), the k-mers at or below tau go into an open-addressing table in LDS
// keyed by hash -- first position (min), last position (max), strand sum, exactly what the sorted runs gave -- and only the
// table is sorted.  Fewer than s distinct hashes (repeats, low complexity): tau x 4 and again, up to "every k-mer"; more
// than the table holds: bisection between the last two thresholds.  A valid k-mer whose hash is the table's EMPTY value
// (2^-64) sets *redo and the host runs the sorting kernel instead.  K = 0: any k <= 32 (byte-wise hashing).
template <int K>
__global__ __launch_bounds__(256) void sketch_fragments_table_kernel(const uint8_t* __restrict__ seq, const int64_t* __restrict__ frag_off,
                                                                     const int32_t* __restrict__ frag_len, int k, int s, int32_t seq_id, int T,
                                                                     wfm_minmer_t* __restrict__ out, int32_t* __restrict__ out_count, int32_t* redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint64_t EMPTY = ~0ull;
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);   // [T]
  uint32_t* mn = reinterpret_cast<uint32_t*>(key + T);  // [T] first position
  uint32_t* mx = mn + T;                                // [T] last position
  int* sm = reinterpret_cast<int*>(mx + T);             // [T] strand sum
  __shared__ int s_cnt, s_over;
  const int f = blockIdx.x, tid = threadIdx.x;
  const uint8_t* p = seq + frag_off[f];
  const int nk = frag_len[f] - k + 1;
  if (nk <= 0) { if (tid == 0) out_count[f] = 0; return; }
  const int cap = T - T / 4;
  constexpr uint64_t TMAX = ~0ull - 1;
  uint64_t tau;
  {
    const double t = (double)(2 * s + 24) / (2.0 * (double)nk);
    tau = t >= 0.999 ? TMAX : (uint64_t)(t * 18446744073709551616.0);
  }
  uint64_t lo = 0, hi = 0;  // the largest threshold known to be too small; the smallest known to overflow (0: none yet)
  int D = 0;
  for (;;) {
    for (int i = tid; i < T; i += blockDim.x) { key[i] = EMPTY; mn[i] = 0xffffffffu; mx[i] = 0u; sm[i] = 0; }
    if (tid == 0) { s_cnt = 0; s_over = 0; }
    __syncthreads();
    for (int i = tid; i < nk; i += blockDim.x) {
      if (*(volatile int*)&s_over) break;
      uint64_t h; int8_t st;
      if (K > 0) kmer_hash_2w<(K > 0 ? K : 9)>(p + i, h, st);
      else {
        uint64_t fw[4], rc[4];
        h = EMPTY; st = 0;
        if (load_kmer(p + i, k, fw, rc)) {
          const uint64_t hf = murmur3_x64_lo(fw, k, 42u), hb = murmur3_x64_lo(rc, k, 42u);
          if (hf != hb) { h = hf < hb ? hf : hb; st = hf < hb ? 1 : -1; }
        }
      }
      if (st == 0) continue;                   // an N inside, or its own reverse complement
      if (h == EMPTY) { *redo = 1; continue; }
      if (h > tau) continue;
      uint32_t slot = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 40) & (uint32_t)(T - 1);
      for (int probe = 0; probe < T; ++probe) {
        const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&key[slot]), (unsigned long long)EMPTY, (unsigned long long)h);
        if (old == EMPTY && atomicAdd(&s_cnt, 1) >= cap) s_over = 1;
        if (old == EMPTY || old == h) {
          atomicMin(&mn[slot], (uint32_t)i); atomicMax(&mx[slot], (uint32_t)i); atomicAdd(&sm[slot], st > 0 ? 1 : -1);
          break;
        }
        slot = (slot + 1) & (uint32_t)(T - 1);
      }
    }
    __syncthreads();
    D = s_cnt;
    const int over = s_over;
    __syncthreads();
    if (over) { hi = tau; tau = lo + (hi - lo) / 2; continue; }
    if (D >= s || tau == TMAX) break;
    lo = tau;
    if (hi) tau = lo + (hi - lo) / 2;
    else tau = tau > TMAX / 4 ? TMAX : tau * 4;
  }
  // the table by hash (bitonic, EMPTY last)
  for (int size = 2; size <= T; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (T >> 1); t += blockDim.x) {
        const int a = 2 * t - (t & (stride - 1)), b = a + stride;
        const bool asc = ((a & size) == 0);
        const uint64_t ka = key[a], kb = key[b];
        if ((ka > kb) == asc && ka != kb) {
          key[a] = kb; key[b] = ka;
          const uint32_t m0 = mn[a], x0 = mx[a]; const int s0 = sm[a];
          mn[a] = mn[b]; mx[a] = mx[b]; sm[a] = sm[b];
          mn[b] = m0; mx[b] = x0; sm[b] = s0;
        }
      }
      __syncthreads();
    }
  }
  const int cnt = min(D, s);
  for (int r = tid; r < cnt; r += blockDim.x) {
    wfm_minmer_t* o = out + (size_t)f * s + r;
    o->hash = key[r]; o->wpos = (int64_t)mn[r]; o->wpos_end = (int64_t)mx[r]; o->seqId = seq_id;
    const int v = sm[r];
    o->strand = (int16_t)(v > 0 ? 1 : (v == 0 ? 0 : -1));  // FWD=1, AMBIG=0, REV=-1
    o->pad_ = 0;
  }
  if (tid == 0) out_count[f] = cnt;
}

}  // namespace wfm

using namespace wfm;

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      wfm_set_error((h), std::string(#call) + ": " + hipGetErrorString(e_));            \
      return WFM_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

namespace {
struct Scoped {
  std::vector<void*> ptrs;
  ~Scoped() { if (!ptrs.empty()) (void)hipDeviceSynchronize(); for (void* p : ptrs) if (p) wfm_dfree_nosync(p); }
  template <typename T> hipError_t alloc(T** p, size_t bytes) {
    hipError_t e = wfm_dmalloc((void**)p, bytes ? bytes : 16);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

// ---- normalised sequences that stay on the device for the length of a map call (wfm_map_sequence_cache) ----
// A map call uploads a target chromosome twice -- once for the sketch of the identity estimate, once for the index -- and a query three times
// (sketch, index when it is a target too, fragments): 5 ms of PCIe and 1.8 ms of normalize_kernel per 249 Mbp each time, 54 of a full-size C4
// rank's 281 ms of index build.  Between wfm_map_sequence_cache(h, 1) and (h, 0) the normalised copy a call makes is kept (up to
// WFM_NORM_CACHE_GB, 16 by default, per process) and found again by the host pointer, the length and the sequence's first and last 32 bytes;
// a consumer's stream waits for the event recorded behind the producer's normalize_kernel.  The host side opens the scope where the files are
// opened and closes it before they are let go (host/capi_map.cpp): a pointer cannot come to mean another sequence in between.
struct NormEntry { int device; const char* seq; int64_t len; unsigned char fp[64]; uint8_t* d_norm; hipEvent_t ready; };
struct NormCache {
  std::mutex mu;
  int scopes = 0;
  size_t bytes = 0;
  std::vector<NormEntry> entries;
};
NormCache& norm_cache() { static NormCache* c = new NormCache; return *c; }
void norm_fingerprint(const char* seq, int64_t len, unsigned char* fp) {
  std::memset(fp, 0, 64);
  const int64_t a = std::min<int64_t>(32, len);
  std::memcpy(fp, seq, (size_t)a);
  std::memcpy(fp + 32, seq + len - a, (size_t)a);
}
const uint8_t* norm_cache_find(int device, const char* seq, int64_t len, hipStream_t consumer) {
  NormCache& c = norm_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.scopes || c.entries.empty()) return nullptr;
  unsigned char fp[64];
  bool have_fp = false;
  for (const NormEntry& e : c.entries) {
    if (e.device != device || e.seq != seq || e.len != len) continue;
    if (!have_fp) { norm_fingerprint(seq, len, fp); have_fp = true; }
    if (std::memcmp(fp, e.fp, 64) != 0) continue;
    if (e.ready && hipStreamWaitEvent(consumer, e.ready, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e.d_norm;
  }
  return nullptr;
}
// the scope is open and the budget has room for another `bytes`
bool norm_cache_wants(size_t bytes) {
  static const size_t budget = (size_t)(getenv("WFM_NORM_CACHE_GB") ? std::max(0, atoi(getenv("WFM_NORM_CACHE_GB"))) : 16) << 30;
  NormCache& c = norm_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  return c.scopes > 0 && c.bytes + bytes <= budget;
}
void norm_cache_add(int device, const char* seq, int64_t len, uint8_t* d_norm, hipStream_t producer) {
  NormEntry e;
  e.device = device; e.seq = seq; e.len = len; e.d_norm = d_norm; e.ready = nullptr;
  norm_fingerprint(seq, len, e.fp);
  if (hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(e.ready, producer) != hipSuccess) {
    (void)hipGetLastError();
    if (e.ready) { (void)hipEventDestroy(e.ready); e.ready = nullptr; }
    (void)hipStreamSynchronize(producer);  // no event: the copy is complete before anybody can find it
  }
  NormCache& c = norm_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  c.bytes += (size_t)len + 64;
  c.entries.push_back(e);
}

// uploads seq and returns a normalised, 64-byte padded device copy (or the copy an earlier call of this map call has left on the device)
int upload_normalised(wfm_handle_t* h, Scoped& sc, const char* seq, int64_t len, uint8_t** d_norm) {
  uint8_t* d_raw = nullptr;
  const size_t padded = (size_t)len + 64;
  hipStream_t st = wfm_stream(h);
  if (const uint8_t* kept = norm_cache_find(wfm_device(h), seq, len, st)) { *d_norm = const_cast<uint8_t*>(kept); return WFM_OK; }
  HIPCHK(h, sc.alloc(&d_raw, padded));
  const bool keep = len >= ((int64_t)1 << 20) && norm_cache_wants(padded);  // (short sequences: the copy is not what their calls cost)
  if (keep) HIPCHK(h, wfm_dmalloc((void**)d_norm, padded));
  else HIPCHK(h, sc.alloc(d_norm, padded));
  struct Kept { uint8_t* p; ~Kept() { if (p) wfm_dfree(p); } } kept_guard{keep ? *d_norm : nullptr};  // (an error below: nobody else has the block yet)
  HIPCHK(h, hipMemsetAsync(*d_norm + len, 'N', padded - (size_t)len, st));  // (only the padding: normalize_kernel writes every base)
  HIPCHK(h, hipMemcpyAsync(d_raw, seq, (size_t)len, hipMemcpyHostToDevice, st));
  const int64_t nthreads = (len + 15) / 16;
  const int blocks = (int)((nthreads + 255) / 256);
  if (blocks > 0) hipLaunchKernelGGL(normalize_kernel, dim3(blocks), dim3(256), 0, st, d_raw, *d_norm, len);
  HIPCHK(h, hipGetLastError());
  if (keep) { norm_cache_add(wfm_device(h), seq, len, *d_norm, st); kept_guard.p = nullptr; }
  return WFM_OK;
}
}  // namespace

int map_sketch_device(wfm_handle_t* h, MapScratch& ms, const char* seq, int64_t seq_len, const int64_t* frag_off,
                      const int32_t* frag_len, size_t n, int k, int s, int32_t seq_id, wfm_minmer_t** d_out_p, int32_t** d_cnt_p) {
  if (k < 1 || k > 32 || s < 1) { wfm_set_error(h, "k must be in 1..32 and s >= 1"); return WFM_E_UNSUPPORTED; }
  int maxk = 1;
  for (size_t i = 0; i < n; ++i) {
    if (frag_off[i] < 0 || frag_len[i] < 0 || frag_off[i] + frag_len[i] > seq_len) { wfm_set_error(h, "fragment out of range"); return WFM_E_ARG; }
    maxk = std::max(maxk, frag_len[i] - k + 1);
  }
  int npow2 = 512;
  while (npow2 < maxk) npow2 <<= 1;
  size_t lds = (size_t)npow2 * 12 + (size_t)s * 4 + 64;
  const bool in_global = lds > 160 * 1024;  // more than 8192 k-mers per fragment: the pairs go to global memory
  if (in_global) lds = (size_t)s * 4 + 64;
  if (in_global && n * (size_t)npow2 * 12 > ((size_t)16 << 30)) { wfm_set_error(h, "fragments too long for this many at once"); return WFM_E_UNSUPPORTED; }
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  Scoped sc;  // the raw / normalised sequence copies are only needed until the sketch kernel has run
  uint8_t* d_norm = nullptr;
  int rc = upload_normalised(h, sc, seq, seq_len, &d_norm);
  if (rc != WFM_OK) return rc;
  int64_t* d_off = nullptr; int32_t* d_len = nullptr; wfm_minmer_t* d_out = nullptr; int32_t* d_cnt = nullptr;
  HIPCHK(h, sc.alloc(&d_off, n * 8));
  HIPCHK(h, sc.alloc(&d_len, n * 4));
  HIPCHK(h, ms.alloc(&d_out, n * (size_t)s));
  HIPCHK(h, ms.alloc(&d_cnt, n));
  hipStream_t st = wfm_stream(h);
  HIPCHK(h, hipMemcpyAsync(d_off, frag_off, n * 8, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_len, frag_len, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(d_out, 0, n * (size_t)s * sizeof(wfm_minmer_t), st));
  if (lds > 64 * 1024) {
    HIPCHK(h, hipFuncSetAttribute((const void*)sketch_fragments_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // the table form while its table (2 x (2 s + 24) slots at least, 20 B each) fits 64 KB of LDS; WFM_SKETCH_TABLE=0: the sort
  const bool table_on = !(getenv("WFM_SKETCH_TABLE") && atoi(getenv("WFM_SKETCH_TABLE")) == 0);
  int T = 256;
  while (T < 2 * (2 * s + 24)) T <<= 1;
  if (table_on && T <= 2048 && maxk >= 1) {
    int32_t* d_redo = nullptr;
    HIPCHK(h, sc.alloc(&d_redo, 4));
    HIPCHK(h, hipMemsetAsync(d_redo, 0, 4, st));
    const size_t tl = (size_t)T * 20;
    switch (k) {
#define WFM_SKT(K) case K: hipLaunchKernelGGL(sketch_fragments_table_kernel<K>, dim3((unsigned)n), dim3(256), tl, st, d_norm, d_off, d_len, k, s, seq_id, T, d_out, d_cnt, d_redo); break;
      WFM_SKT(9) WFM_SKT(10) WFM_SKT(11) WFM_SKT(12) WFM_SKT(13) WFM_SKT(14) WFM_SKT(15) WFM_SKT(16)
      WFM_SKT(17) WFM_SKT(18) WFM_SKT(19) WFM_SKT(20) WFM_SKT(21) WFM_SKT(22) WFM_SKT(23) WFM_SKT(24)
#undef WFM_SKT
      default: hipLaunchKernelGGL(sketch_fragments_table_kernel<0>, dim3((unsigned)n), dim3(256), tl, st, d_norm, d_off, d_len, k, s, seq_id, T, d_out, d_cnt, d_redo);
    }
    HIPCHK(h, hipGetLastError());
    int32_t redo = 0;
    HIPCHK(h, hipMemcpyAsync(&redo, d_redo, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));  // (d_norm / d_off / d_len are released on return)
    if (!redo) { *d_out_p = d_out; *d_cnt_p = d_cnt; return WFM_OK; }
    HIPCHK(h, hipMemsetAsync(d_out, 0, n * (size_t)s * sizeof(wfm_minmer_t), st));
  }
  uint64_t* d_gkey = nullptr; uint32_t* d_gpv = nullptr;
  if (in_global) {
    HIPCHK(h, sc.alloc(&d_gkey, n * (size_t)npow2 * 8));
    HIPCHK(h, sc.alloc(&d_gpv, n * (size_t)npow2 * 4));
  }
  hipLaunchKernelGGL(sketch_fragments_kernel, dim3((unsigned)n), dim3(256), lds, st, d_norm, d_off, d_len, k, s, seq_id, npow2, d_out, d_cnt, d_gkey, d_gpv);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(st));  // d_norm / d_off / d_len are released on return
  *d_out_p = d_out; *d_cnt_p = d_cnt;
  return WFM_OK;
}

// ---- hashed sequence kept on the device; host threads fetch slices concurrently (minmers.cpp) ----
int map_hash_sequence_device(wfm_handle_t* h, const char* seq, int64_t len, int k, MapHashedSeq* out) {
  out->d_norm = nullptr; out->d_hash = nullptr; out->d_strand = nullptr; out->len = len; out->nk = len - k + 1; out->device = wfm_device(h);
  if (out->nk <= 0) return WFM_OK;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  const size_t padded = (size_t)len + 64;
  uint8_t* d_raw = nullptr;
  auto fail = [&](hipError_t e, const char* what) {
    if (d_raw) (void)wfm_dfree(d_raw);
    map_hashed_free(out);
    wfm_set_error(h, std::string(what) + ": " + hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? WFM_E_NOMEM : WFM_E_HIP;
  };
  hipError_t e;
  if ((e = wfm_dmalloc((void**)&d_raw, padded)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = wfm_dmalloc((void**)&out->d_norm, padded)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = wfm_dmalloc((void**)&out->d_hash, (size_t)out->nk * 8)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = wfm_dmalloc((void**)&out->d_strand, (size_t)out->nk)) != hipSuccess) return fail(e, "hipMalloc");
  if ((e = hipMemsetAsync(out->d_norm + len, 'N', padded - (size_t)len, st)) != hipSuccess) return fail(e, "hipMemsetAsync");
  if ((e = hipMemcpyAsync(d_raw, seq, (size_t)len, hipMemcpyHostToDevice, st)) != hipSuccess) return fail(e, "hipMemcpyAsync");
  const int64_t nthreads = (len + 15) / 16;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, d_raw, out->d_norm, len);
  launch_kmer_hash(out->d_norm, out->nk, k, out->d_hash, out->d_strand, st);
  if ((e = hipGetLastError()) != hipSuccess) return fail(e, "kernel launch");
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "hipStreamSynchronize");
  wfm_dfree_nosync(d_raw);  // (its only user, normalize_kernel on st, has finished)
  return WFM_OK;
}

void map_hash_work_free(MapHashWork* wk) {
  (void)hipSetDevice(wk->device);
  if (wk->d_raw) (void)wfm_dfree(wk->d_raw);
  if (wk->d_norm) (void)wfm_dfree(wk->d_norm);
  if (wk->d_hash) (void)wfm_dfree(wk->d_hash);
  if (wk->d_strand) (void)wfm_dfree(wk->d_strand);
  *wk = MapHashWork();
}

int map_hash_sequence_into(wfm_handle_t* h, MapHashWork* wk, const char* seq, int64_t len, int k, MapHashedSeq* out) {
  *out = MapHashedSeq();
  out->len = len; out->nk = len - k + 1; out->device = wfm_device(h); out->borrowed = true;
  if (out->nk <= 0) return WFM_OK;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  if ((size_t)len > wk->cap) {
    map_hash_work_free(wk);
    wk->device = wfm_device(h);
    const size_t cap = (size_t)len + (size_t)len / 8;  // a little room: chromosomes come in similar sizes
    hipError_t e = wfm_dmalloc((void**)&wk->d_raw, cap + 64);
    if (e == hipSuccess) e = wfm_dmalloc((void**)&wk->d_norm, cap + 64);
    if (e == hipSuccess) e = wfm_dmalloc((void**)&wk->d_hash, cap * 8);
    if (e == hipSuccess) e = wfm_dmalloc((void**)&wk->d_strand, cap);
    if (e != hipSuccess) {
      map_hash_work_free(wk);
      wfm_set_error(h, std::string("hipMalloc: ") + hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? WFM_E_NOMEM : WFM_E_HIP;
    }
    wk->cap = cap;
  }
  // (the normalised copy the identity estimate's sketch left on the device, if this map call keeps them: no upload, no normalize_kernel)
  const uint8_t* d_norm = norm_cache_find(wfm_device(h), seq, len, st);
  if (!d_norm) {
    HIPCHK(h, hipMemsetAsync(wk->d_norm + len, 'N', 64, st));  // the hash kernel reads whole words past the end
    HIPCHK(h, hipMemcpyAsync(wk->d_raw, seq, (size_t)len, hipMemcpyHostToDevice, st));
    const int64_t nthreads = (len + 15) / 16;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, wk->d_raw, wk->d_norm, len);
    d_norm = wk->d_norm;
  }
  launch_kmer_hash(d_norm, out->nk, k, wk->d_hash, wk->d_strand, st);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(st));
  out->d_norm = const_cast<uint8_t*>(d_norm); out->d_hash = wk->d_hash; out->d_strand = wk->d_strand;
  return WFM_OK;
}

void map_hashed_free(MapHashedSeq* s) {
  if (s->borrowed) { s->d_norm = nullptr; s->d_hash = nullptr; s->d_strand = nullptr; return; }
  (void)hipSetDevice(s->device);
  if (s->d_norm) (void)wfm_dfree(s->d_norm);
  if (s->d_hash) (void)wfm_dfree(s->d_hash);
  if (s->d_strand) (void)wfm_dfree(s->d_strand);
  s->d_norm = nullptr; s->d_hash = nullptr; s->d_strand = nullptr;
}

// k-mer starts [from, to) and bases [base_from, base_to), into the host arrays at the same indices.
// Fastest into host memory that is already resident (a copy into fresh pages is bound by page faults).
int map_hashed_fetch(const MapHashedSeq* s, int64_t from, int64_t to, int64_t base_from, int64_t base_to, uint64_t* hash, int8_t* strand,
                     char* norm) {
  if (hipSetDevice(s->device) != hipSuccess) return WFM_E_HIP;
  if (to > from) {
    if (hipMemcpy(hash + from, s->d_hash + from, (size_t)(to - from) * 8, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpy(strand + from, s->d_strand + from, (size_t)(to - from), hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
  }
  if (base_to > base_from && hipMemcpy(norm + base_from, s->d_norm + base_from, (size_t)(base_to - base_from), hipMemcpyDeviceToHost) != hipSuccess)
    return WFM_E_HIP;
  return WFM_OK;
}

namespace {
void map_stage_destroy(void* p) {
  MapStage* st = static_cast<MapStage*>(p);
  (void)hipSetDevice(st->device);
  for (hipEvent_t e : st->ev) if (e) (void)hipEventDestroy(e);
  if (st->stream) (void)hipStreamDestroy(st->stream);
  if (st->base) (void)hipHostFree(st->base);
  delete st;
}
}  // namespace

int map_stage_acquire(wfm_handle_t* h, size_t slot_bytes, int nslots, MapStage** out) {
  MapStage* st = static_cast<MapStage*>(wfm_attachment(h));
  if (st && st->slot_bytes >= slot_bytes && st->nslots >= nslots) { *out = st; return WFM_OK; }
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  st = new MapStage();
  st->device = wfm_device(h);
  st->nslots = nslots;
  st->slot_bytes = (slot_bytes + 4095) & ~(size_t)4095;
  hipError_t e = hipHostMalloc((void**)&st->base, st->slot_bytes * (size_t)nslots, hipHostMallocDefault);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking);
  st->ev.assign((size_t)nslots, nullptr);
  for (int i = 0; i < nslots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&st->ev[(size_t)i], hipEventDisableTiming);
  if (e != hipSuccess) {
    map_stage_destroy(st);
    wfm_set_error(h, std::string("pinned staging ring: ") + hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? WFM_E_NOMEM : WFM_E_HIP;
  }
  wfm_set_attachment(h, st, map_stage_destroy);  // replaces (and frees) a smaller ring
  *out = st;
  return WFM_OK;
}

int map_stage_copy(MapStage* st, int slot, const MapHashedSeq* s, int64_t from, int64_t to, int64_t base_from, int64_t base_to) {
  const size_t nk = (size_t)std::max<int64_t>(0, to - from), nb = (size_t)std::max<int64_t>(0, base_to - base_from);
  if (slot < 0 || slot >= st->nslots || nk * 9 + nb > st->slot_bytes) return WFM_E_ARG;
  if (hipSetDevice(st->device) != hipSuccess) return WFM_E_HIP;
  char* dst = st->slot(slot);
  if (nk) {
    if (hipMemcpyAsync(dst, s->d_hash + from, nk * 8, hipMemcpyDeviceToHost, st->stream) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpyAsync(dst + nk * 8, s->d_strand + from, nk, hipMemcpyDeviceToHost, st->stream) != hipSuccess) return WFM_E_HIP;
  }
  if (nb && hipMemcpyAsync(dst + nk * 9, s->d_norm + base_from, nb, hipMemcpyDeviceToHost, st->stream) != hipSuccess) return WFM_E_HIP;
  return hipEventRecord(st->ev[(size_t)slot], st->stream) == hipSuccess ? WFM_OK : WFM_E_HIP;
}

int map_stage_wait(MapStage* st, int slot) {
  if (hipSetDevice(st->device) != hipSuccess) return WFM_E_HIP;
  return hipEventSynchronize(st->ev[(size_t)slot]) == hipSuccess ? WFM_OK : WFM_E_HIP;
}

int map_hashed_fetch_packed(const MapHashedSeq* s, int64_t from, int64_t to, int64_t base_from, int64_t base_to, char* dst) {
  if (hipSetDevice(s->device) != hipSuccess) return WFM_E_HIP;
  const size_t nk = (size_t)std::max<int64_t>(0, to - from), nb = (size_t)std::max<int64_t>(0, base_to - base_from);
  if (nk) {
    if (hipMemcpy(dst, s->d_hash + from, nk * 8, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
    if (hipMemcpy(dst + nk * 8, s->d_strand + from, nk, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
  }
  if (nb && hipMemcpy(dst + nk * 9, s->d_norm + base_from, nb, hipMemcpyDeviceToHost) != hipSuccess) return WFM_E_HIP;
  return WFM_OK;
}

// wfm_hash_kmers; norm_out (optional, len bytes) receives the upper-cased / N-masked sequence the
// hashes were computed from, for callers that go on working on the host (minmers.cpp)
int wfm_hash_kmers_norm(wfm_handle_t* h, const char* seq, int64_t len, int k, uint64_t* hash, int8_t* strand, char* norm_out) {
  if (!h || !seq || !hash || !strand || len < 0) return WFM_E_ARG;
  if (k < 1 || k > 32) { wfm_set_error(h, "k must be in 1..32"); return WFM_E_UNSUPPORTED; }
  const int64_t nk = len - k + 1;
  if (nk <= 0) return WFM_OK;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  Scoped sc;
  uint8_t* d_norm = nullptr;
  int rc = upload_normalised(h, sc, seq, len, &d_norm);
  if (rc != WFM_OK) return rc;
  uint64_t* d_hash = nullptr; int8_t* d_st = nullptr;
  HIPCHK(h, sc.alloc(&d_hash, (size_t)nk * 8));
  HIPCHK(h, sc.alloc(&d_st, (size_t)nk));
  hipStream_t st = wfm_stream(h);
  launch_kmer_hash(d_norm, nk, k, d_hash, d_st, st);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(hash, d_hash, (size_t)nk * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(strand, d_st, (size_t)nk, hipMemcpyDeviceToHost, st));
  if (norm_out) HIPCHK(h, hipMemcpyAsync(norm_out, d_norm, (size_t)len, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return WFM_OK;
}

extern "C" {

int wfm_map_sequence_cache(wfm_handle_t* h, int open) {
  if (!h) return WFM_E_ARG;
  NormCache& c = norm_cache();
  std::vector<NormEntry> gone;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    if (open) { ++c.scopes; return WFM_OK; }
    if (c.scopes > 0 && --c.scopes == 0) { gone.swap(c.entries); c.bytes = 0; }
  }
  for (NormEntry& e : gone) {  // (wfm_dfree waits for the device: no kernel still reads the block)
    (void)hipSetDevice(e.device);
    if (e.ready) (void)hipEventDestroy(e.ready);
    wfm_dfree(e.d_norm);
  }
  if (!gone.empty()) (void)hipSetDevice(wfm_device(h));
  return WFM_OK;
}

int wfm_hash_kmers(wfm_handle_t* h, const char* seq, int64_t len, int k, uint64_t* hash, int8_t* strand) {
  return wfm_hash_kmers_norm(h, seq, len, k, hash, strand, nullptr);
}

// Bottom-`sketch_size` MinHash of one sequence as StreamingMinHash keeps it (streamingMinHash.hpp:90-100):
// the smallest canonical hashes WITH multiplicity of all k-mers free of non-ACGT bases whose two
// strands hash differently (map_stats.hpp:569-616).  An ambiguous base among the first k bases
// blanks k-mers 0..k-1 (the counter is armed with k there, :574-580).  One hashing pass + one
// device radix sort.
int64_t wfm_minhash_sketch(wfm_handle_t* h, const char* seq, int64_t len, int k, int sketch_size, uint64_t* out) {
  if (!h || !seq || len < 0 || sketch_size < 1 || !out) return WFM_E_ARG;
  if (k < 1 || k > 32) { wfm_set_error(h, "k must be in 1..32"); return WFM_E_UNSUPPORTED; }
  const int64_t nk = len - k + 1;
  if (nk <= 0) return 0;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  const bool dbg = getenv("WFM_DEBUG") != nullptr;
  const auto tq0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  Scoped sc;
  uint8_t* d_norm = nullptr;
  int rc = upload_normalised(h, sc, seq, len, &d_norm);
  if (rc != WFM_OK) return rc;
  hipStream_t st = wfm_stream(h);
  bool head_ambiguous = false;
  for (int j = 0; j < k && j < len; ++j) {
    char c = seq[j];
    if (c > 96 && c < 123) c -= 32;
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') { head_ambiguous = true; break; }
  }
  const int64_t n = std::min<int64_t>(sketch_size, nk);
  // ---- one pass (round 5): hash and keep what is at most tau.  The sketch is the n smallest hashes with multiplicity; a canonical hash is the
  // smaller of two uniform values, so a fraction t of the range holds ~2 t of the k-mers: tau for about 8 n of them.  The two-pass form below
  // (hash everything: 2 GB per chromosome, select, sort) remains for short sequences, other k and the rare sequence with fewer than n below tau.
  static const bool fused_on = !(getenv("WFM_MINHASH_FUSED") && atoi(getenv("WFM_MINHASH_FUSED")) == 0);
  if (fused_on && nk > ((int64_t)1 << 20) && n * 64 < nk) {
    const long double t = 4.0L * (long double)n / (long double)nk;
    const uint64_t tau = (uint64_t)(t * 18446744073709551616.0L);
    const unsigned long long cap = (unsigned long long)n * 64;
    uint64_t *d_sel = nullptr, *d_srt = nullptr; unsigned long long* d_count = nullptr;
    HIPCHK(h, sc.alloc(&d_sel, (size_t)cap * 8));
    HIPCHK(h, sc.alloc(&d_srt, (size_t)cap * 8));
    HIPCHK(h, sc.alloc(&d_count, sizeof(unsigned long long)));
    HIPCHK(h, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st));
    if (dbg) { HIPCHK(h, hipStreamSynchronize(st)); fprintf(stderr, "[wfm] minhash_sketch of %lld bases: allocations + upload + normalise %.1f ms", (long long)len, since(tq0)); }
    const auto tf1 = std::chrono::steady_clock::now();
    if (launch_kmer_hash_select(d_norm, nk, k, head_ambiguous ? std::min<int64_t>(k, nk) : 0, tau, d_sel, d_count, cap, st)) {
      HIPCHK(h, hipGetLastError());
      unsigned long long m = 0;
      HIPCHK(h, hipMemcpyAsync(&m, d_count, sizeof(m), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      if (dbg) fprintf(stderr, ", hashing + selection in one pass %.1f ms (%llu at most tau)", since(tf1), m);
      if ((int64_t)m >= n && m <= cap) {
        const auto tf2 = std::chrono::steady_clock::now();
        size_t tmp = 0;
        HIPCHK(h, rocprim::radix_sort_keys(nullptr, tmp, d_sel, d_srt, (size_t)m, 0, 64, st));
        char* d_tmp = nullptr;
        HIPCHK(h, sc.alloc(&d_tmp, tmp));
        HIPCHK(h, rocprim::radix_sort_keys(d_tmp, tmp, d_sel, d_srt, (size_t)m, 0, 64, st));
        HIPCHK(h, hipMemcpyAsync(out, d_srt, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (dbg) fprintf(stderr, ", sort + download %.1f ms\n", since(tf2));
        int64_t valid = n;
        while (valid > 0 && out[valid - 1] == ~0ull) --valid;
        return valid;
      }
      if (dbg) fprintf(stderr, " -- outside [n, 64 n]: the two-pass form\n");
    } else if (dbg) fprintf(stderr, " -- no one-pass form for k = %d\n", k);
  }
  uint64_t *d_hash = nullptr, *d_sorted = nullptr; int8_t* d_st = nullptr;
  HIPCHK(h, sc.alloc(&d_hash, (size_t)nk * 8));
  HIPCHK(h, sc.alloc(&d_sorted, (size_t)nk * 8));
  HIPCHK(h, sc.alloc(&d_st, (size_t)nk));
  if (dbg) { HIPCHK(h, hipStreamSynchronize(st)); fprintf(stderr, "[wfm] minhash_sketch of %lld bases (two passes): allocations + upload + normalise %.1f ms", (long long)len, since(tq0)); }
  const auto tq1 = std::chrono::steady_clock::now();
  launch_kmer_hash(d_norm, nk, k, d_hash, d_st, st);
  HIPCHK(h, hipGetLastError());
  if (dbg) { HIPCHK(h, hipStreamSynchronize(st)); fprintf(stderr, ", hashing %.1f ms", since(tq1)); }
  const auto tq2 = std::chrono::steady_clock::now();
  if (head_ambiguous) HIPCHK(h, hipMemsetAsync(d_hash, 0xff, (size_t)std::min<int64_t>(k, nk) * 8, st));
  // The sketch is the n smallest hashes (with multiplicity): sorting a chromosome's 2.5e8 hashes for 4096 of them
  // is most of the ANI estimate's time.  Hashes at most tau are selected first -- tau set so that about 8 n of
  // them are expected (a canonical hash is the smaller of two uniform values: a fraction t of the range holds
  // ~2t of the k-mers) -- and only those are sorted; if fewer than n turn up, everything is sorted as before.
  bool done = false;
  if (nk > ((int64_t)1 << 20) && n * 64 < nk) {
    const long double t = 4.0L * (long double)n / (long double)nk;
    const uint64_t tau = (uint64_t)(t * 18446744073709551616.0L);
    unsigned long long* d_count = nullptr;
    HIPCHK(h, sc.alloc(&d_count, sizeof(unsigned long long)));
    auto below = [tau] __device__(const uint64_t& v) { return v <= tau; };
    size_t stmp = 0;
    HIPCHK(h, rocprim::select(nullptr, stmp, d_hash, d_sorted, d_count, (size_t)nk, below, st));
    char* d_stmp = nullptr;
    HIPCHK(h, sc.alloc(&d_stmp, stmp));
    HIPCHK(h, rocprim::select(d_stmp, stmp, d_hash, d_sorted, d_count, (size_t)nk, below, st));
    unsigned long long m = 0;
    HIPCHK(h, hipMemcpyAsync(&m, d_count, sizeof(m), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if ((int64_t)m >= n) {
      // d_sorted holds the m selected hashes; sort them back into d_hash (no longer needed)
      size_t tmp = 0;
      HIPCHK(h, rocprim::radix_sort_keys(nullptr, tmp, d_sorted, d_hash, (size_t)m, 0, 64, st));
      char* d_tmp = nullptr;
      HIPCHK(h, sc.alloc(&d_tmp, tmp));
      HIPCHK(h, rocprim::radix_sort_keys(d_tmp, tmp, d_sorted, d_hash, (size_t)m, 0, 64, st));
      HIPCHK(h, hipMemcpyAsync(out, d_hash, (size_t)n * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      done = true;
    }
  }
  if (!done) {
    size_t tmp = 0;
    HIPCHK(h, rocprim::radix_sort_keys(nullptr, tmp, d_hash, d_sorted, (size_t)nk, 0, 64, st));
    char* d_tmp = nullptr;
    HIPCHK(h, sc.alloc(&d_tmp, tmp));
    HIPCHK(h, rocprim::radix_sort_keys(d_tmp, tmp, d_hash, d_sorted, (size_t)nk, 0, 64, st));
    HIPCHK(h, hipMemcpyAsync(out, d_sorted, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
  }
  if (dbg) fprintf(stderr, ", select + sort + download %.1f ms (%s)\n", since(tq2), done ? "threshold" : "full sort");
  int64_t valid = n;
  while (valid > 0 && out[valid - 1] == ~0ull) --valid;  // invalid k-mers sort last
  return valid;
}

int wfm_sketch_fragments(wfm_handle_t* h, const char* seq, int64_t seq_len, const int64_t* frag_off,
                         const int32_t* frag_len, size_t n, int k, int s, int32_t seq_id,
                         wfm_minmer_t* out, int32_t* out_count) {
  if (!h || !seq || (n && (!frag_off || !frag_len || !out || !out_count)) || seq_len < 0) return WFM_E_ARG;
  if (n == 0) return WFM_OK;
  MapScratch sc;
  wfm_minmer_t* d_out = nullptr; int32_t* d_cnt = nullptr;
  const int rc = map_sketch_device(h, sc, seq, seq_len, frag_off, frag_len, n, k, s, seq_id, &d_out, &d_cnt);
  if (rc != WFM_OK) return rc;
  hipStream_t st = wfm_stream(h);
  HIPCHK(h, hipMemcpyAsync(out, d_out, n * (size_t)s * sizeof(wfm_minmer_t), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(out_count, d_cnt, n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return WFM_OK;
}

}  // extern "C"
