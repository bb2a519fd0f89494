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
#include <limits>
#include <string>
#include <vector>

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

}  // namespace

extern "C" int64_t wfm_map_fragments(wfm_handle_t* h, const wfm_index_t* ix, const char* seq, int64_t seq_len, const int64_t* frag_off,
                                     const int32_t* frag_seq_id, int64_t nfrag, const wfm_map_params_t* prm, wfm_mapping_t* out,
                                     int32_t* out_frag, int64_t cap) {
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
  if (n_out > 0 && out && out_frag && cap > 0) {
    const size_t n_copy = (size_t)std::min<int64_t>(n_out, cap);
    HIPCHK(h, hipMemcpyAsync(out, d_out, n_copy * sizeof(wfm_mapping_t), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(out_frag, d_ofrag, n_copy * 4, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(h, hipStreamSynchronize(st));
  return n_out;
}
