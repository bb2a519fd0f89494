// map_device.h -- device-pointer entry points of the map stages, shared by the per-stage C ABI
// wrappers and the fused wfm_map_fragments path (map_fragments.hip).
#ifndef WFM_MAP_DEVICE_H_
#define WFM_MAP_DEVICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "wfa_handle.h"
#include "dev_cache.h"

// device allocations that live until the owning call returns
struct MapScratch {
  std::vector<void*> p;
  MapScratch() = default;
  MapScratch(const MapScratch&) = delete;
  MapScratch& operator=(const MapScratch&) = delete;
  ~MapScratch() { if (!p.empty()) (void)hipDeviceSynchronize(); for (void* q : p) if (q) wfm_dfree_nosync(q); }
  template <typename T> hipError_t alloc(T** out, size_t n) {
    hipError_t e = wfm_dmalloc((void**)out, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) p.push_back(*out);
    return e;
  }
};

// sketchSequence of n fragments of one host buffer; results stay on the device (n x s minmers + counts)
int map_sketch_device(wfm_handle_t* h, MapScratch& sc, const char* seq, int64_t seq_len, const int64_t* frag_off,
                      const int32_t* frag_len, size_t n, int k, int s, int32_t seq_id, wfm_minmer_t** d_out, int32_t** d_cnt);

// L1 on device-resident sketches; prm's tables are host pointers.  *d_cands is owned by sc.
int map_l1_device(wfm_handle_t* h, MapScratch& sc, const wfm_index_t* ix, const wfm_minmer_t* d_q, const int32_t* d_qcount,
                  const int32_t* d_qseq, const int32_t* d_qlen, const uint8_t* d_active, int64_t nfrag, int s,
                  const wfm_l1_params_t* prm, wfm_l1_candidate_t** d_cands, int64_t* ncand);

// L2 on device-resident sketches and candidates.  *d_out / *d_frag are owned by sc.
int map_l2_device(wfm_handle_t* h, MapScratch& sc, const wfm_index_t* ix, const wfm_minmer_t* d_q, const int32_t* d_qcount,
                  const int32_t* d_qlen, const uint8_t* d_kc, int64_t nfrag, int s, const wfm_l1_candidate_t* d_cands,
                  int64_t ncand, const wfm_l2_params_t* prm, wfm_mapping_t** d_out, int32_t** d_frag, int64_t* n_out);

// The index stage on minmer intervals that are already on the device (map_index.hip); d_minmers stays the caller's.
int map_index_build_device(wfm_handle_t* h, const wfm_minmer_t* d_minmers, int64_t n, double max_kmer_freq, wfm_index_t** out);

// One sequence normalised and hashed on the device, left there so that host threads can pull the
// slices they work on in parallel (each through its own per-thread stream).
struct MapHashedSeq {
  uint8_t* d_norm = nullptr;   // upper-cased / N-masked bases
  uint64_t* d_hash = nullptr;  // canonical hash per k-mer start (~0 = invalid)
  int8_t* d_strand = nullptr;  // +1 / -1 / 0 (contains N or palindromic)
  int64_t len = 0, nk = 0;
  int device = 0;
  bool borrowed = false;       // the arrays belong to a MapHashWork: valid until it hashes the next sequence
};
int map_hash_sequence_device(wfm_handle_t* h, const char* seq, int64_t len, int k, MapHashedSeq* out);
// grow-only device buffers for hashing one sequence after the other without allocating each time
struct MapHashWork {
  uint8_t *d_raw = nullptr, *d_norm = nullptr;
  uint64_t* d_hash = nullptr;
  int8_t* d_strand = nullptr;
  size_t cap = 0;  // bases
  int device = 0;
};
int map_hash_sequence_into(wfm_handle_t* h, MapHashWork* wk, const char* seq, int64_t len, int k, MapHashedSeq* out);
void map_hash_work_free(MapHashWork* wk);
void map_hashed_free(MapHashedSeq* s);
int map_hashed_fetch(const MapHashedSeq* s, int64_t from, int64_t to, int64_t base_from, int64_t base_to, uint64_t* hash, int8_t* strand,
                     char* norm);

// Pinned staging ring for streaming hashed slices to host workers (minmers.cpp): a slot receives
//   uint64 hash[to-from] | int8 strand[to-from] | char norm[base_to-base_from]
// by asynchronous copies on the ring's own stream; map_stage_wait blocks the calling thread until the
// slot's copies have landed.  The ring is kept with the handle and reused by later calls.
struct MapStage {
  int device = 0;
  int nslots = 0;
  size_t slot_bytes = 0;
  char* base = nullptr;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;
  char* slot(int i) const { return base + (size_t)i * slot_bytes; }
};
inline size_t map_stage_bytes(int64_t nkmers, int64_t nbases) { return (size_t)nkmers * 9 + (size_t)nbases; }
int map_stage_acquire(wfm_handle_t* h, size_t slot_bytes, int nslots, MapStage** out);
int map_stage_copy(MapStage* st, int slot, const MapHashedSeq* s, int64_t from, int64_t to, int64_t base_from, int64_t base_to);
int map_stage_wait(MapStage* st, int slot);
// the same layout, synchronously, into ordinary memory (the rare replays)
int map_hashed_fetch_packed(const MapHashedSeq* s, int64_t from, int64_t to, int64_t base_from, int64_t base_to, char* dst);

// The k-mers of a hashed sequence the host winnowing has to see (map_prefilter.hip), ascending positions.
struct MapSparseSeq {
  uint32_t* d_pos = nullptr;   // k-mer start
  uint64_t* d_hash = nullptr;
  int8_t* d_strand = nullptr;  // +1 / -1
  int64_t m = 0;
  int device = 0;
};
// The threshold that lets about c x s of the W k-mers of a window through.  A canonical hash is the smaller of
// two uniform 64-bit values, so a fraction t of the hash range holds 1 - (1-t)^2 of the k-mers.
inline uint64_t map_prefilter_tau(double c_factor, int s, int64_t W) {
  const double share = c_factor * (double)s / (double)std::max<int64_t>(1, W);  // of the k-mers
  if (!(share < 1.0)) return ~0ull;
  const long double t = 1.0L - sqrtl(1.0L - (long double)share);
  return (uint64_t)(t * 18446744073709551616.0L);
}
// grow-only scratch of the thinning passes (optional: without it every call allocates and frees its own)
struct MapThinWork {
  struct Buf { void* p = nullptr; size_t bytes = 0; };
  Buf a, b, ck, ck2, cp, cp2, tmp;
  int device = 0;
};
void map_thin_work_free(MapThinWork* wk);
// Device blocks that come and go once per sequence (the kept k-mers, a sequence's records) are taken from and returned to
// a small pool: hipMalloc / hipFree wait for the whole device, and with the winnowing on a thread and stream of its own
// every such call would make the two threads wait for each other's kernels.  map_dev_pool_trim frees what is pooled
// (add_minmers_core calls it before it returns: the memory belongs to whoever runs next).
void* map_dev_pool_get(int device, size_t bytes);
void map_dev_pool_put(int device, void* p);
void map_dev_pool_trim();
// W = k-mers per window (w - k + 1), s = sketch size, tau = hash threshold
int map_prefilter_device(wfm_handle_t* h, const MapHashedSeq* q, int64_t W, int s, uint64_t tau, MapSparseSeq* out, MapThinWork* wk = nullptr);
void map_sparse_free(MapSparseSeq* s);
// out[i] = number of kept k-mers with position < query[i] (host arrays)
int map_sparse_lower_bound(wfm_handle_t* h, const MapSparseSeq* s, const int64_t* query, int nq, int64_t* out, hipStream_t stream = nullptr);
// kept k-mers [c0, c1): uint64 hash[mc] | uint32 pos[mc] | int8 strand[mc], into a ring slot / ordinary memory
int map_stage_copy_sparse(MapStage* st, int slot, const MapSparseSeq* s, int64_t c0, int64_t c1);
int map_sparse_fetch_packed(const MapSparseSeq* s, int64_t c0, int64_t c1, char* dst);

// The winnowing of a thinned stream on the device (map_winnow.hip): one wave per speculative chunk.
struct MapWinnowWork {  // grow-only device buffers, reused from sequence to sequence
  struct Buf { void* p = nullptr; size_t bytes = 0; };
  Buf chunks, recs, count, st_begin, st_end, wp_end, flags, off, out, todo;
};
void map_winnow_work_free(MapWinnowWork* wk);
struct MapWinnowInfo { int chunks = 0, bad_chunks = 0, rerun_chunks = 0, replays = 0, resolve_rounds = 0; uint32_t why = 0; int64_t records = 0; };
// WFM_OK: *d_out (inside wk, valid until the next call) holds *n_out raw records in emission order, interval starts
// resolved; 1: this sequence is not for the device (info->why), the caller winnows it on the host; < 0: error
int map_winnow_sparse_device(wfm_handle_t* h, const MapSparseSeq* sp, int64_t len, int k, int w, int s, int32_t seq_id, int64_t chunk_len,
                             MapWinnowWork* wk, wfm_minmer_t** d_out, int64_t* n_out, MapWinnowInfo* info, hipStream_t stream = nullptr);
// The closing steps of addMinmers on the device (map_finish.hip): pieces of at most w windows, strand signs, std::sort's
// order by (wpos, wpos_end) -- ties as libstdc++'s introsort leaves them -- and de-duplication.
struct MapFinishWork {  // grow-only device buffers
  struct Buf { void* p = nullptr; size_t bytes = 0; };
  Buf ns, np, os, op, R, key, idx, key2, idx2, A, B, seg[4], small_, heap, counts, tiles, tile_cnt, tile0, info, tmp, out;
};
void map_finish_work_free(MapFinishWork* wk);
struct MapFinishInfo { int64_t laid_out = 0, records = 0; int levels = 0, heap_ranges = 0; };
int map_finish_records_device(wfm_handle_t* h, const wfm_minmer_t* d_raw, int64_t n_raw, int w, MapFinishWork* wk, wfm_minmer_t** d_out, int64_t* n_out,
                              MapFinishInfo* info, hipStream_t stream = nullptr);
void map_sortlike_model(std::vector<std::pair<uint64_t, uint32_t>>& v);  // the same arrangement computed on the host (CPU test-suite)

// the kernel's control flow and capacities on plain host arrays (CPU test-suite); -1 = the device would hand the sequence back
int64_t map_winnow_model(const uint32_t* pos, const uint64_t* hash, const int8_t* strand, int64_t m, int64_t len, int k, int w, int s, int32_t seq_id,
                         int64_t chunk_len, std::vector<wfm_minmer_t>* out, uint32_t* why, int force_replay = 0, int* replays_out = nullptr);
#endif
