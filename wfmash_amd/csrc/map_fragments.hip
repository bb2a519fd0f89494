// map_fragments.hip -- the fused per-batch map path: sketch -> L1 -> L2 with everything between
// the stages resident in HBM (SURVEY 8b "wfm_map_fragments"; the reference's per-fragment
// Map::mapSingleQueryFrag, src/map/include/computeMap.hpp:875-938, run over a whole batch).
//
// Only two small host round trips remain: the per-fragment (count, largest hash) pair, from which
// the host evaluates Q.kmerComplexity in the reference's long double / double / float mix
// (mappingCore.hpp:72-74 -- there is no long double on the device), and the stage totals that
// size the next stage's arenas.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/wfmash_hip.h"
#include "map_device.h"
#include "wfa_handle.h"

namespace {

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      wfm_set_error((h), std::string(#call) + ": " + hipGetErrorString(e_));            \
      return WFM_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

__global__ void last_hash_kernel(const wfm_minmer_t* q, const int32_t* cnt, int s, int64_t nfrag, uint64_t* last) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nfrag) return;
  const int c = cnt[f];
  last[f] = c > 0 ? q[f * s + c - 1].hash : 0;
}

__global__ void fill_i32_kernel(int32_t* p, int32_t v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- f3, the first step of the filters on the device: the order chaining begins with ----
// mergeMappingsInRange[WithChains] (mappingFilter.hpp:402-421, :593-612) begins by sorting a query's mappings by (target, strand, query
// position, target position).  The mappings are still on the device when L2 ends: two stable radix sorts (low key: the two positions, high key:
// query, target, strand) give that order for a whole batch, a pass over neighbours says whether any two mappings of a query share a key (the
// reference's std::sort leaves such ties in an order of its own: the host sorts then, as before).  The host gets the permutation with the
// mappings, builds each query's vector in chaining order straight from it, checks that the keys ascend strictly (host/map_filter.cpp) and
// skips its own sort and the permutation of the 28-byte records.
__global__ void chain_keys_kernel(const wfm_mapping_t* __restrict__ m, const int32_t* __restrict__ frag, const int32_t* __restrict__ frag_first, int w,
                                  uint64_t* __restrict__ klo, uint64_t* __restrict__ khi, uint32_t* __restrict__ idx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const wfm_mapping_t a = m[i];
  const int32_t f = frag[i], f0 = frag_first[f];
  const uint32_t qpos = a.queryStartPos + (uint32_t)((f - f0) * w);  // what the caller adds (computeMap.hpp:124-128), in the same 32 bits
  klo[i] = ((uint64_t)qpos << 32) | a.refStartPos;
  khi[i] = ((uint64_t)(uint32_t)f0 << 32) | ((uint64_t)(a.refSeqId & 0x7fffffffu) << 1) | ((a.flags & 1) ? 0u : 1u);  // (reverse strand -1 sorts before forward +1)
  idx[i] = (uint32_t)i;
}
__global__ void gather_u64_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// the low keys (positions) in the final order, made again from the mappings (the first sort's output order is gone after the second)
__global__ void gather_keylo_kernel(const wfm_mapping_t* __restrict__ m, const int32_t* __restrict__ frag, const int32_t* __restrict__ frag_first, int w,
                                    const uint32_t* __restrict__ perm, uint64_t* __restrict__ klo, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = perm[i];
  const int32_t f = frag[j], f0 = frag_first[f];
  klo[i] = ((uint64_t)(m[j].queryStartPos + (uint32_t)((f - f0) * w)) << 32) | m[j].refStartPos;
}
__global__ void chain_ties_kernel(const uint64_t* __restrict__ klo, const uint64_t* __restrict__ khi, const wfm_mapping_t* __restrict__ m, int* __restrict__ ties, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (m[i].refSeqId > 0x7fffffffu) *ties = 1;  // (an id the key does not hold)
  if (i + 1 < n && khi[i] == khi[i + 1] && klo[i] == klo[i + 1]) *ties = 1;
}

}  // namespace

static int64_t map_fragments_impl(wfm_handle_t* h, const wfm_index_t* ix, const char* seq, int64_t seq_len, const int64_t* frag_off,
                                  const int32_t* frag_seq_id, int64_t nfrag, const wfm_map_params_t* prm, wfm_mapping_t* out,
                                  int32_t* out_frag, int64_t cap, const int32_t* frag_first, uint32_t* out_perm) {
  if (!h || !ix || !prm || nfrag < 0 || seq_len < 0 || (nfrag && (!seq || !frag_off || !frag_seq_id))) return WFM_E_ARG;
  const wfm_l1_params_t& p1 = prm->l1;
  const wfm_l2_params_t& p2 = prm->l2;
  if (p1.window_length != p2.window_length || p1.sketch_size != p2.sketch_size || p1.sketch_size < 1) return WFM_E_ARG;
  if (!p1.ref_group || !p1.min_hits_by_qsketch || !p1.sketch_cutoffs || p1.n_cutoffs < 1 || !p2.keep_table || !p2.ident_table || !p2.cutoff_j)
    return WFM_E_ARG;
  if (nfrag == 0) return 0;
  for (int64_t f = 0; f < nfrag; ++f)
    if (frag_seq_id[f] < 0 || frag_seq_id[f] >= p1.n_seq) { wfm_set_error(h, "wfm_map_fragments: query seqId out of range"); return WFM_E_ARG; }
  const int w = p1.window_length, s = p1.sketch_size, k = prm->kmer_size;
  HIPCHK(h, hipSetDevice(wfm_device(h)));
  hipStream_t st = wfm_stream(h);
  MapScratch sc;
  std::vector<int32_t> flen((size_t)nfrag, w);
  wfm_minmer_t* d_q = nullptr; int32_t* d_cnt = nullptr;
  int rc = map_sketch_device(h, sc, seq, seq_len, frag_off, flen.data(), (size_t)nfrag, k, s, 0, &d_q, &d_cnt);
  if (rc != WFM_OK) return rc;
  // Q.kmerComplexity on the host
  uint64_t* d_last = nullptr; int32_t *d_qseq = nullptr, *d_qlen = nullptr; uint8_t *d_act = nullptr, *d_kc = nullptr;
  if (sc.alloc(&d_last, nfrag) != hipSuccess || sc.alloc(&d_qseq, nfrag) != hipSuccess || sc.alloc(&d_qlen, nfrag) != hipSuccess ||
      sc.alloc(&d_act, nfrag) != hipSuccess || sc.alloc(&d_kc, nfrag) != hipSuccess) { wfm_set_error(h, "out of device memory (map)"); return WFM_E_NOMEM; }
  const dim3 g((unsigned)((nfrag + 255) / 256)), b(256);
  hipLaunchKernelGGL(last_hash_kernel, g, b, 0, st, d_q, d_cnt, s, nfrag, d_last);
  hipLaunchKernelGGL(fill_i32_kernel, g, b, 0, st, d_qlen, (int32_t)w, nfrag);
  std::vector<uint64_t> last((size_t)nfrag);
  std::vector<int32_t> cnt((size_t)nfrag);
  HIPCHK(h, hipMemcpyAsync(last.data(), d_last, (size_t)nfrag * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nfrag * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(d_qseq, frag_seq_id, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipStreamSynchronize(st));
  std::vector<uint8_t> act((size_t)nfrag), kc((size_t)nfrag);
  for (int64_t f = 0; f < nfrag; ++f) {
    if (cnt[f] == 0) { act[f] = 0; kc[f] = 0; continue; }
    const double max_hash_01 = (long double)(last[f]) / (long double)std::numeric_limits<uint64_t>::max();  // host pass: x87 long double
    const float complexity = (double(cnt[f]) / max_hash_01) / ((w - k + 1) * 2);
    act[f] = !(complexity < prm->kmer_complexity_threshold);
    kc[f] = (uint8_t)(int)roundf(complexity * 100.0f);  // MappingResult::setKmerComplexity
  }
  HIPCHK(h, hipMemcpyAsync(d_act, act.data(), (size_t)nfrag, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(d_kc, kc.data(), (size_t)nfrag, hipMemcpyHostToDevice, st));
  wfm_l1_candidate_t* d_cands = nullptr; int64_t ncand = 0;
  rc = map_l1_device(h, sc, ix, d_q, d_cnt, d_qseq, d_qlen, d_act, nfrag, s, &p1, &d_cands, &ncand);
  if (rc != WFM_OK) return rc;
  wfm_mapping_t* d_out = nullptr; int32_t* d_ofrag = nullptr; int64_t n_out = 0;
  rc = map_l2_device(h, sc, ix, d_q, d_cnt, d_qlen, d_kc, nfrag, s, d_cands, ncand, &p2, &d_out, &d_ofrag, &n_out);
  if (rc != WFM_OK) return rc;
  int ties = 1;
  if (n_out > 0 && out && out_frag && cap > 0) {
    const size_t n_copy = (size_t)std::min<int64_t>(n_out, cap);
    HIPCHK(h, hipMemcpyAsync(out, d_out, n_copy * sizeof(wfm_mapping_t), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(out_frag, d_ofrag, n_copy * 4, hipMemcpyDeviceToHost, st));
    if (out_perm && frag_first && n_out <= cap && n_out < ((int64_t)1 << 32)) {
      // the chaining order of the batch (see chain_keys_kernel); failures here only cost the shortcut
      uint64_t *d_klo = nullptr, *d_khi = nullptr, *d_k2 = nullptr;
      uint32_t *d_i0 = nullptr, *d_i1 = nullptr;
      int32_t* d_ff = nullptr; int* d_ties = nullptr;
      const size_t n = (size_t)n_out;
      if (sc.alloc(&d_klo, n) == hipSuccess && sc.alloc(&d_khi, n) == hipSuccess && sc.alloc(&d_k2, n) == hipSuccess && sc.alloc(&d_i0, n) == hipSuccess &&
          sc.alloc(&d_i1, n) == hipSuccess && sc.alloc(&d_ff, (size_t)nfrag) == hipSuccess && sc.alloc(&d_ties, 1) == hipSuccess) {
        HIPCHK(h, hipMemcpyAsync(d_ff, frag_first, (size_t)nfrag * 4, hipMemcpyHostToDevice, st));
        HIPCHK(h, hipMemsetAsync(d_ties, 0, sizeof(int), st));
        const dim3 gm((unsigned)((n + 255) / 256));
        hipLaunchKernelGGL(chain_keys_kernel, gm, dim3(256), 0, st, d_out, d_ofrag, d_ff, w, d_klo, d_khi, d_i0, (int64_t)n);
        size_t tmp = 0;
        HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp, d_klo, d_k2, d_i0, d_i1, n, 0, 64, st));
        void* d_tmp = nullptr;
        if (sc.alloc((char**)&d_tmp, tmp) == hipSuccess) {
          HIPCHK(h, rocprim::radix_sort_pairs(d_tmp, tmp, d_klo, d_k2, d_i0, d_i1, n, 0, 64, st));  // by the positions: d_k2 (sorted low keys), d_i1
          // the high keys in that order, then the stable sort by them (values: the indices; the low keys travel as a second pass of the same sort)
          uint64_t* d_khs = d_klo;  // (d_klo is free now)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(gather_u64_kernel), gm, dim3(256), 0, st, d_khi, d_i1, d_khs, (int64_t)n);
          size_t tmp2 = 0;
          HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp2, d_khs, d_khi, d_i1, d_i0, n, 0, 64, st));
          void* d_tmp2 = d_tmp;
          if (tmp2 <= tmp || sc.alloc((char**)&d_tmp2, tmp2) == hipSuccess) {
            HIPCHK(h, rocprim::radix_sort_pairs(d_tmp2, tmp2, d_khs, d_khi, d_i1, d_i0, n, 0, 64, st));  // d_khi: sorted high keys, d_i0: the permutation
            hipLaunchKernelGGL(HIP_KERNEL_NAME(gather_keylo_kernel), gm, dim3(256), 0, st, d_out, d_ofrag, d_ff, w, d_i0, d_k2, (int64_t)n);
            hipLaunchKernelGGL(chain_ties_kernel, gm, dim3(256), 0, st, d_k2, d_khi, d_out, d_ties, (int64_t)n);
            HIPCHK(h, hipMemcpyAsync(out_perm, d_i0, n * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipMemcpyAsync(&ties, d_ties, sizeof(int), hipMemcpyDeviceToHost, st));
          }
        }
      }
      (void)hipGetLastError();
    }
  }
  HIPCHK(h, hipStreamSynchronize(st));
  if (out_perm && n_out > 0 && ties) out_perm[0] = 0xffffffffu;  // no shortcut for this batch: the host sorts
  return n_out;
}

extern "C" int64_t wfm_map_fragments(wfm_handle_t* h, const wfm_index_t* ix, const char* seq, int64_t seq_len, const int64_t* frag_off,
                                     const int32_t* frag_seq_id, int64_t nfrag, const wfm_map_params_t* prm, wfm_mapping_t* out,
                                     int32_t* out_frag, int64_t cap) {
  return map_fragments_impl(h, ix, seq, seq_len, frag_off, frag_seq_id, nfrag, prm, out, out_frag, cap, nullptr, nullptr);
}

extern "C" int64_t wfm_map_fragments_ordered(wfm_handle_t* h, const wfm_index_t* ix, const char* seq, int64_t seq_len, const int64_t* frag_off,
                                             const int32_t* frag_seq_id, int64_t nfrag, const wfm_map_params_t* prm, wfm_mapping_t* out,
                                             int32_t* out_frag, int64_t cap, const int32_t* frag_first, uint32_t* out_perm) {
  return map_fragments_impl(h, ix, seq, seq_len, frag_off, frag_seq_id, nfrag, prm, out, out_frag, cap, frag_first, out_perm);
}
