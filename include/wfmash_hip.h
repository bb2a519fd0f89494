/*
 * include/wfmash_hip.h -- C ABI of libwfmash_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the wfmash hot path.  Every entry point below replaces
 * one seam of the reference (file:line relative to waveygang/wfmash):
 *
 *   align path
 *     wfm_align_batch   <- wfa::WFAlignerGapAffine2Pieces::alignEnd2End /
 *                          alignEndsFree + getAlignment as used at
 *                          src/common/wflign/src/wflign.cpp:136-148 (main BiWFA,
 *                          MemoryUltralow), :280-309 (head patch, MemoryMed),
 *                          :368-401 (tail patch) and
 *                          src/common/wflign/src/wflign_alignment.cpp:665-678
 *                          (wflign_edit_cigar_copy).  One call aligns a whole
 *                          batch of mapping records (the batch seam above
 *                          Aligner::processAlignment,
 *                          src/align/include/computeAlignments.hpp:661).
 *   map path
 *     wfm_hash_kmers        <- CommonFunc::getHash, src/map/include/commonFunc.hpp:173-182
 *                              (MurmurHash3_x64_128 seed 42, src/common/murmur3.h:226-302)
 *     wfm_sketch_fragments  <- CommonFunc::sketchSequence, commonFunc.hpp:218-323
 *                              via MappingCore::getSeedHits, mappingCore.hpp:62-76
 *     wfm_add_minmers[_multi] <- CommonFunc::addMinmers, commonFunc.hpp:440-708
 *     wfm_prefilter_kmers   <- (no counterpart: the device-side thinning of addMinmers' input stream)
 *     wfm_finish_records    <- the closing steps of addMinmers, commonFunc.hpp:660-706 (pieces of w windows, strand signs,
 *                              std::sort by (wpos, wpos_end) with the library's tie order, std::unique)
 *     wfm_index_build       <- Sketch::build (index stage), winSketch.hpp:266-429
 *     wfm_index_build_sequences <- Sketch::build as a whole, winSketch.hpp:175-457
 *     wfm_index_replicate   <- (the one Sketch all mapping threads share, computeMap.hpp:431-484: one copy per GPU)
 *     wfm_index_upload / wfm_index_download <- Sketch::readIndex / writeIndex (device side), winSketch.hpp:569-866
 *     wfm_map_l1            <- getSeedIntervalPoints + computeL1CandidateRegions, mappingCore.hpp:82-301
 *     wfm_map_l2            <- computeL2MappedRegions + SlideMapper + doL2Mapping,
 *                              mappingCore.hpp:307-442, slidingMap.hpp:28-212, computeMap.hpp:989-1061
 *     wfm_map_fragments     <- Map::mapSingleQueryFrag for a whole batch, computeMap.hpp:875-938
 *     wfm_minhash_sketch    <- StreamingMinHash over one sequence (ANI estimate), map_stats.hpp:569-616
 *   (file-level drivers of both phases: include/wfmash_host.h)
 *
 * All pointers are HOST memory unless stated; the library owns every device
 * buffer.  No torch types.  A handle is bound to one GPU and is not
 * thread-safe; use one handle per host thread / per rank.
 * Every function returns 0 on success or a negative WFM_E_* code; there is no
 * CPU fallback: without a visible gfx950 device wfm_create fails.
 */
#ifndef WFMASH_HIP_H_
#define WFMASH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFM_OK              0
#define WFM_E_NODEVICE     (-1)
#define WFM_E_HIP          (-2)   /* a HIP runtime call failed; see wfm_last_error */
#define WFM_E_ARG          (-3)
#define WFM_E_NOMEM        (-4)
#define WFM_E_UNSUPPORTED  (-5)   /* e.g. penalties whose score scope exceeds the ring */
#define WFM_E_ARENA        (-6)   /* caller's ops arena too small */

/* per-problem status (wfm_result_t.status); 0 mirrors WF_STATUS_ALG_COMPLETED
 * tested at wflign.cpp:150,307,399 */
#define WFM_ST_OK            0
#define WFM_ST_UNREACHABLE (-300)
#define WFM_ST_OOM         (-200)

#define WFMASH_HIP_VERSION "wfmash-hip-0.3"   /* SAM @PG VN:, `wfmash-hip --version` */

typedef struct wfm_handle wfm_handle_t;

/* wflign_penalties_t (wflign_alignment.hpp:21) minus match (always 0) */
typedef struct { int32_t x, o1, e1, o2, e2; } wfm_penalties_t;

/* alignment mode = the WFAligner memory model / entry point the reference uses */
#define WFM_MODE_END2END_BIWFA 0   /* alignEnd2End, MemoryUltralow  (wflign.cpp:136-148) */
#define WFM_MODE_ENDSFREE      1   /* alignEndsFree, MemoryMed      (wflign.cpp:280-305,368-397) */
#define WFM_MODE_END2END_UNI   2   /* alignEnd2End, full backtrace (MemoryHigh)            */

typedef struct {
  const char* pattern;  int32_t plen;     /* wfmash passes pattern = target */
  const char* text;     int32_t tlen;     /* wfmash passes text    = query  */
  int32_t mode;                           /* WFM_MODE_*                      */
  int32_t pattern_begin_free, pattern_end_free;   /* ENDSFREE only */
  int32_t text_begin_free, text_end_free;
  int32_t score_hint;                     /* END2END_BIWFA only; 0 = none.  A guess of an upper bound of the alignment's
                                           * score (e.g. the cost of the end gaps the caller's padding implies plus a
                                           * divergence allowance).  The wavefronts are then only computed where an
                                           * alignment of at most that score can pass; a guess that turns out too small
                                           * costs a second run without it, never a different result. */
  int32_t pad_;
} wfm_problem_t;

typedef struct {
  int32_t  status;     /* WFM_ST_*                                             */
  int32_t  score;      /* gap-affine-2p penalty of the returned alignment      */
  uint64_t ops_off;    /* offset of this problem's op string in ops_arena (wfm_align_batch_rle: index of its first run) */
  uint32_t ops_len;    /* number of ops over {M,X,I,D}; I = text-only, D = pattern-only */
  uint32_t n_runs;     /* number of run-length encoded CIGAR runs             */
  uint64_t cells;      /* (score,diagonal) cells computed on the device        */
} wfm_result_t;

typedef struct {
  uint64_t cells;            /* cells computed in the last batch                  */
  uint64_t bytes_algorithmic;/* 48*cells + sum(plen+tlen)         */
  double   ms_kernels;       /* device time of all WFA kernels (hipEvent)         */
  double   ms_breakpoint;    /* device time of the BiWFA breakpoint kernels       */
  double   ms_base;          /* device time of the base / backtrace kernels       */
  double   ms_total;         /* host wall time of the whole call                  */
  uint32_t levels;           /* BiWFA recursion levels executed                   */
  uint32_t bp_jobs, base_jobs;
  uint32_t bp_launches, base_launches;
  uint64_t cells_bp;         /* cells computed by wfa_bp_kernel launches          */
  uint64_t cells_base;       /* cells computed by wfa_base_kernel launches        */
  uint64_t cells_tile;       /* part of cells_bp computed by wfa_tile_kernel      */
  double   ms_tile;          /* part of ms_breakpoint spent in wfa_tile_kernel    */
  uint32_t tile_launches, tile_tasks;
  double   ms_tile_busy;     /* time during which at least one tile kernel launch was running: equals ms_tile unless
                              * the two halves of a batch ran their tile kernels side by side on two streams */
  uint32_t streams;          /* parts of the batch that ran side by side, each on its own stream */
  uint32_t pad_;
  uint64_t cells_tile_unique;/* cells_tile without the block a job computes twice: the full block in which its wavefronts
                              * meet is thrown away and run again up to the meeting point only */
  double   ms_bp_busy;       /* as ms_tile_busy, for the wfa_bp_kernel launches (ms_breakpoint - ms_tile summed over streams) */
  double   ms_base_busy;     /* as ms_tile_busy, for the wfa_base_kernel launches */
  double   ms_any_busy;      /* time during which any of the three kernels was running */
  uint32_t p2_launches, p2_jobs; /* phase 2 from rows computed ahead (tile kernel + wfa_p2_* kernels): launch sets, jobs */
  uint32_t p2_more, p2_again; /* of those jobs, the ones wfa_bp_kernel had to finish step by step; further rounds of rows computed ahead (jobs x rounds) */
} wfm_stats_t;

int  wfm_device_count(void);   /* usable HIP devices of this node (0 without a GPU) */
int  wfm_create(int device, wfm_handle_t** out);
void wfm_destroy(wfm_handle_t* h);
const char* wfm_last_error(const wfm_handle_t* h);
int  wfm_device_name(const wfm_handle_t* h, char* buf, size_t buflen);

/* Upper bound of the ops_arena bytes wfm_align_batch needs for these problems
 * (sum of plen+tlen+1). */
size_t wfm_align_arena_bytes(const wfm_problem_t* problems, size_t n);

/* Align n problems.  out[i].ops_off/ops_len locate the op string of problem i
 * inside ops_arena (not NUL-terminated).  Returns the number of problems whose
 * status != 0, or a negative WFM_E_* code if the call itself failed. */
int  wfm_align_batch(wfm_handle_t* h, const wfm_penalties_t* pen,
                     const wfm_problem_t* problems, size_t n,
                     wfm_result_t* out, char* ops_arena, size_t arena_bytes);

/* The same with run-length output, the form the align driver consumes: the reference's own pipeline compresses the op
 * string at once (compress_cigar, wflign.cpp:183-208) and works on runs from there on (erosion scan, merge, swizzle and
 * write_alignment_paf: wflign.cpp:174-231,241-454), so nothing on that path needs one byte per base; wfm_align_batch's
 * expanded form stays for getAlignment(char**, int*) (wflign_alignment.cpp:671-677).
 * *runs receives a buffer of 32-bit runs owned by the caller (release with wfm_free_runs), run = (length << 2) | op
 * with op 0 = M, 1 = X, 2 = I (text only), 3 = D (pattern only); adjacent runs of one problem never share an op.
 * out[i].ops_off = index of problem i's first run in *runs, out[i].n_runs their number, out[i].ops_len the number of
 * ops they spell.  Returns as wfm_align_batch. */
#define WFM_RUN_LEN(r) ((uint32_t)(r) >> 2)
#define WFM_RUN_OP(r)  ((uint32_t)(r) & 3u)
int  wfm_align_batch_rle(wfm_handle_t* h, const wfm_penalties_t* pen,
                         const wfm_problem_t* problems, size_t n,
                         wfm_result_t* out, uint32_t** runs, size_t* n_runs_total);
void wfm_free_runs(uint32_t* runs);

/* Upper bounds of the END2END scores of n problems without aligning them: the cost of one valid global alignment per
 * problem, found by a greedy walk on the device (one wave per problem: extend along the diagonal, at a difference try a
 * substitution and the indels up to 31 bases side by side; wfmash_amd/csrc/wfa_kernels.hip, wfa_bound_kernel).  out[i] >= the
 * optimal gap-affine-2p score of problem i, or -1 where the walk gave up (divergent sequences, structural differences,
 * sequences under 256 bases).  wfm_align_batch runs this itself for its long BiWFA problems and cuts their wavefronts to
 * what an alignment of at most that score can touch; the entry point exists for callers that want the bound and for tests. */
int  wfm_score_bounds(wfm_handle_t* h, const wfm_penalties_t* pen, const wfm_problem_t* problems, size_t n, int32_t* out);

/* Self-test of the cross-lane primitives the packed tile kernel leans on (DPP wave shifts, wfmash_amd/csrc/wfa_tile2.hip):
 * out128[lane] = the value 1000 + (lane - 1) taken from the previous lane (lane 0: the kernel's NULL, -2^30),
 * out128[64 + lane] = 1000 + (lane + 1) from the next lane (lane 63: NULL). */
int  wfm_selftest_dpp(wfm_handle_t* h, int32_t* out128);
/* Self-test of the device arenas' growth policy: capacities (in 4-byte elements) after a first request of n0 elements, after a request of one element
 * more than that capacity (must at least double it), and after a request that fits (must not allocate). */
int  wfm_selftest_arena_growth(wfm_handle_t* h, size_t n0, size_t* out3);

/* Same, but sequences are already resident in device memory (the timed region
 * of bench.py starts here): d_seqs is a device pointer, offsets index into it.
 * Sequences must be laid out by wfm_upload_sequences. */
typedef struct wfm_seqset wfm_seqset_t;
int  wfm_upload_sequences(wfm_handle_t* h, const wfm_problem_t* problems, size_t n, wfm_seqset_t** out);
void wfm_free_sequences(wfm_handle_t* h, wfm_seqset_t* s);
int  wfm_align_resident(wfm_handle_t* h, const wfm_penalties_t* pen, wfm_seqset_t* s,
                        wfm_result_t* out, char* ops_arena, size_t arena_bytes);
int  wfm_align_resident_rle(wfm_handle_t* h, const wfm_penalties_t* pen, wfm_seqset_t* s,
                            wfm_result_t* out, uint32_t** runs, size_t* n_runs_total);

/* How many other align calls the caller keeps in flight on this handle's DEVICE while a call on this handle runs (the align
 * driver's workers, each on a handle of its own: wfmash_amd/host/aligner.cpp).  0, the default: none -- a batch is then cut
 * into up to three parts that run side by side on streams of their own, because the levels of a batch of near-identical
 * records are chains of short launches (a block of 100 scores takes its 0.13 ms however few workgroups it has) and the device
 * is only filled by several chains at once; with other calls beside it a batch of hundreds of hinted records stays one part
 * (three chains on a device are what its hardware queues run in parallel; gpurun_out/r5f_ab.log). */
void wfm_set_concurrent_calls(wfm_handle_t* h, int other_calls);

/* Which of the rarer paths the problems of the handle's LAST align call took, one word of WFM_PF_* bits per problem (a
 * diagnostic channel: the parity tests and bench.py draw their samples from it, so that the records whose root ran again,
 * whose patches went to a second or third score budget, or which ran on the byte kernels are all checked against the oracle
 * instead of the one in fifty a uniform sample holds).  out may be NULL; returns the number of problems of that call. */
#define WFM_PF_ROOT_AGAIN   1u   /* the root ran past its score hint / out of its narrow ring and was run again        */
#define WFM_PF_JOB_AGAIN    2u   /* a BiWFA child ran out of its narrow ring and was run again on a full one          */
#define WFM_PF_BASE_RETRY   4u   /* a leaf / ends-free patch overflowed its first score budget                        */
#define WFM_PF_BASE_RETRY2  8u   /* ... and its second (1020): the third attempt runs to the all-gap bound            */
#define WFM_PF_BYTE_KERNEL 16u   /* an N or a soft-masked base: the byte kernels instead of the 2-bit packed ones     */
#define WFM_PF_P2_ROUNDS   32u   /* an overlap walk went past the first round of rows computed ahead                  */
#define WFM_PF_RING_KERNEL 64u   /* a leaf / patch ran on the global-memory ring kernel (other penalties, an N; rows beyond 2048 diagonals until round 6) */
#define WFM_PF_BASE_TILES 128u   /* a patch with rows beyond 2048 diagonals ran as tiles of the register kernel (its third attempt)   */
size_t wfm_get_problem_flags(const wfm_handle_t* h, uint32_t* out, size_t n);

/* Device blocks of both paths -- the map path's work buffers, the align path's arenas and a batch's sequence buffers, also
 * those of a handle that has been destroyed -- come from one heap per device and go back to it (wfmash_amd/csrc/dev_cache.h: an
 * address range reserved at the first wfm_create of the device, 1 GB chunks mapped behind what is there, WFM_POOL_GB -- 24 -- at once;
 * memory a process has never had costs ~30 ms per GB on this driver beyond its first ~30 GB, profiles/r6_vmm_fresh.md).  This hands
 * the free chunks at every heap's end (and the cached blocks under a megabyte) back to the driver and returns the bytes released. */
size_t wfm_trim_device_cache(void);

int  wfm_get_stats(const wfm_handle_t* h, wfm_stats_t* out);
/* The intervals during which a kernel of the handle's last align call was running, merged, as (start, end) pairs in ms on a
 * clock all handles of one device share: a caller that keeps several calls in flight on handles of their own (the align
 * driver does) merges them to get the time the device was really busy.  Returns their number (may exceed cap). */
size_t wfm_get_busy_intervals(const wfm_handle_t* h, double* start_end_ms, size_t cap);

/* ---- map path (see header comment for the reference functions) ---- */

/* Canonical k-mer hashes of every k-mer start 0..len-k of seq (upper-cased,
 * non-ACGT treated as N as makeUpperCaseAndValidDNA, commonFunc.hpp:132-142).
 * hash[i]   = min(getHash(fwd), getHash(revcomp)) ; strand[i] = +1 fwd < rev,
 * -1 rev < fwd, 0 palindromic or contains N (hash[i] undefined = UINT64_MAX). */
int  wfm_hash_kmers(wfm_handle_t* h, const char* seq, int64_t len, int k,
                    uint64_t* hash, int8_t* strand);

/* skch::MinmerInfo (base_types.hpp:28-35), 32 bytes */
typedef struct {
  uint64_t hash;
  int64_t  wpos;
  int64_t  wpos_end;
  int32_t  seqId;
  int16_t  strand;
  int16_t  pad_;
} wfm_minmer_t;

/* sketchSequence over n fragments of one buffer: fragment f = seq[frag_off[f] ..
 * frag_off[f]+frag_len[f]).  For each fragment the bottom-s distinct canonical
 * hashes ascending (commonFunc.hpp:218-323).  out holds n*s entries;
 * out_count[f] <= s. */
int  wfm_sketch_fragments(wfm_handle_t* h, const char* seq, int64_t seq_len,
                          const int64_t* frag_off, const int32_t* frag_len, size_t n,
                          int k, int s, int32_t seq_id,
                          wfm_minmer_t* out, int32_t* out_count);

/* skch::IntervalPoint (base_types.hpp:63-76), 24 bytes: endpoints of minmer intervals */
typedef struct {
  int64_t  pos;
  uint64_t hash;
  int32_t  seqId;
  int8_t   side;      /* OPEN = 1, CLOSE = -1 (base_types.hpp:123-127) */
  int8_t   pad_[3];
} wfm_interval_point_t;

/* Device-resident reference index: Sketch::build's index stage (winSketch.hpp:266-429).
 * `minmers` = the minmer intervals of all target sequences concatenated in seqId order (the
 * output of wfm_add_minmers per sequence).  max_kmer_freq as -F (parse_args.hpp:734-737;
 * <= 1: fraction of windows, > 1: absolute count; default 0.0002). */
typedef struct wfm_index wfm_index_t;
typedef struct {
  int64_t  n_windows;   /* minmer intervals given ("windows") */
  int64_t  n_kept;      /* size of minmerIndex after the frequency filter */
  int64_t  n_unique;    /* unique hashes in the position lookup */
  int64_t  n_points;    /* interval points */
  uint64_t threshold;   /* count_threshold actually used */
  int64_t  filtered;    /* intervals dropped by the frequency filter */
  int32_t  adjusted;    /* 1 if the over-filtering safety check raised the threshold (winSketch.hpp:326-349) */
  int32_t  pad_;
} wfm_index_info_t;
int  wfm_index_build(wfm_handle_t* h, const wfm_minmer_t* minmers, int64_t n, double max_kmer_freq, wfm_index_t** out);
void wfm_index_free(wfm_handle_t* h, wfm_index_t* ix);
int  wfm_index_info(const wfm_index_t* ix, wfm_index_info_t* out);
/* Copies the index back to the host (any pointer may be NULL): unique hashes ascending,
 * n_unique+1 offsets into points, the points, and minmerIndex. */
int  wfm_index_download(wfm_handle_t* h, const wfm_index_t* ix, uint64_t* uhash, int64_t* poff,
                        wfm_interval_point_t* points, wfm_minmer_t* minmers);

/* The index on another GPU of the same node (the index is replicated, read-only, on every device that maps:
 * computeMap.hpp:431-484 shares one Sketch between all worker threads).  Device-to-device copies of the four
 * arrays; *out belongs to dst (wfm_index_free(dst, *out)).  src == dst gives a second copy on the same device. */
int  wfm_index_replicate(wfm_handle_t* src, const wfm_index_t* ix, wfm_handle_t* dst, wfm_index_t** out);

/* The inverse of wfm_index_download: a device index from the structures of an index file (`-I`;
 * Sketch::readIndex, winSketch.hpp:834-866, host/index_file.cpp reads the file).  uhash ascending,
 * poff[n_unique + 1] offsets into points, minmers = minmerIndex sorted by (seqId, wpos). */
int  wfm_index_upload(wfm_handle_t* h, const uint64_t* uhash, const int64_t* poff, int64_t n_unique,
                      const wfm_interval_point_t* points, const wfm_minmer_t* minmers, int64_t n_kept,
                      wfm_index_t** out);

/* addMinmers (commonFunc.hpp:440-708): winnowed minmer intervals [wpos, wpos_end) of one target
 * sequence, sorted by (wpos, wpos_end), spans chunked to <= w.  This single-sequence form hashes on
 * the GPU and winnows on the calling host thread (the threads == 1 path; it is what the tests use
 * as the one-stream form).  The production path is wfm_add_minmers_multi / wfm_index_build_sequences:
 * hashing, thinning, winnowing (one wave per speculative chunk) and the closing sort all on the device.
 * Returns the number of intervals (which may exceed cap; only cap are written) or a negative WFM_E_* code. */
int64_t wfm_add_minmers(wfm_handle_t* h, const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                        wfm_minmer_t* out, int64_t cap);

/* wfm_add_minmers for nseq sequences (the reference: one sequence per ThreadPool worker,
 * winSketch.hpp:200-239).  With threads > 1 every stage runs on the device: hashing, thinning of the
 * k-mer stream, winnowing (map_winnow.hip: one wave per speculative chunk, chunks of many sequences in
 * one launch) and the closing std::sort order (map_finish.hip); a sequence the device hands back (an N
 * among its first k-mers that the reference does not notice, a refill anomaly) is winnowed by the
 * `threads` host workers.  out receives the intervals of all sequences concatenated
 * in input order, counts[i] (optional) the number of sequence i.  Returns the total (may exceed
 * cap; only cap are written) or a WFM_E_* code. */
int64_t wfm_add_minmers_multi(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids,
                              int64_t nseq, int k, int w, int s, int threads, wfm_minmer_t* out, int64_t cap, int64_t* counts);

/* Sketch::build in one call (winSketch.hpp:175-457): wfm_add_minmers_multi followed by wfm_index_build, with the
 * minmer intervals going from the host workers straight to the device (no host array of all intervals).  The
 * index equals wfm_index_build(wfm_add_minmers_multi(...)).  *n_windows (optional) receives the number of
 * intervals; when it is 0 no index is made and *out stays NULL. */
int wfm_index_build_sequences(wfm_handle_t* h, const char* const* seqs, const int64_t* lens, const int32_t* seq_ids,
                              int64_t nseq, int k, int w, int s, int threads, double max_kmer_freq,
                              wfm_index_t** out, int64_t* n_windows);

/* The k-mers of a sequence that wfm_add_minmers_multi lets its host workers see: valid k-mers whose hash
 * is under the threshold that lets c_factor * s of a window's w-k+1 k-mers through, plus every valid k-mer of a window that may hold
 * fewer than s distinct such hashes (wfmash_amd/csrc/map_prefilter.hip states the rule and why winnowing
 * the kept k-mers alone reproduces addMinmers).  pos / hash / strand receive up to cap kept k-mers in
 * ascending position; returns their number (may exceed cap) or a WFM_E_* code. */
int64_t wfm_prefilter_kmers(wfm_handle_t* h, const char* seq, int64_t len, int k, int w, int s, double c_factor,
                            uint32_t* pos, uint64_t* hash, int8_t* strand, int64_t cap);

/* The closing steps of addMinmers (commonFunc.hpp:660-706) on the device, on raw interval records in the order the
 * winnower emitted them (what wfm_add_minmers[_multi] runs after its winnowing kernel; wfmash_amd/csrc/map_finish.hip): records
 * with wpos == wpos_end or a negative bound go, strand tallies become signs, records of more than w windows are cut into
 * pieces, everything is ordered by (wpos, wpos_end) -- records that tie in the order libstdc++'s std::sort leaves them --
 * and consecutive records with equal (wpos, hash) are reduced to the first.  Returns the number of records (may exceed
 * cap; only cap are written) or a WFM_E_* code; *levels (optional) = recursion depth the order took, *heap_ranges
 * (optional) = ranges that spent introsort's depth budget and were heap-sorted by the library on the host. */
int64_t wfm_finish_records(wfm_handle_t* h, const wfm_minmer_t* raw, int64_t n, int w, wfm_minmer_t* out, int64_t cap,
                           int32_t* levels, int32_t* heap_ranges);

/* MinHash of one whole sequence for the ANI estimate (estimate_identity_for_groups,
 * src/map/include/map_stats.hpp:325-822; StreamingMinHash, streamingMinHash.hpp:35-135): the
 * sketch_size smallest canonical k-mer hashes, duplicates included, ascending.  Returns how many
 * were written (< sketch_size for short or N-rich sequences) or a WFM_E_* code. */
int64_t wfm_minhash_sketch(wfm_handle_t* h, const char* seq, int64_t len, int k, int sketch_size, uint64_t* out);
/* Between wfm_map_sequence_cache(h, 1) and wfm_map_sequence_cache(h, 0) the normalised device copy of every sequence of a megabase and more that
 * wfm_minhash_sketch / wfm_sketch_fragments / wfm_hash_kmers upload stays on its device (WFM_NORM_CACHE_GB, 16, at most per process), and a later
 * call of those or of wfm_index_build_sequences with the same host pointer, length and first / last 32 bytes uses it instead of uploading and
 * normalising again -- the identity estimate and the index of one map call read the same chromosomes (mashmap's main.cpp:72-128 then
 * computeMap.hpp:405-484: two passes over the files there too).  The caller keeps the host memory alive and unchanged while the scope is open;
 * scopes nest, the last close releases the copies.  wfmh_map_paf opens one around its whole call. */
int wfm_map_sequence_cache(wfm_handle_t* h, int open);

/* L1 stage: getSeedIntervalPoints + computeL1CandidateRegions + doL1Mapping's group loop
 * (mappingCore.hpp:82-301, computeMap.hpp:945-984) for a batch of query fragments against a
 * device-resident index.  One candidate = skch::L1_candidateLocus_t (base_types.hpp:212-224)
 * plus the fragment it belongs to; candidates come out grouped by fragment, in the reference's
 * order within a fragment. */
typedef struct {
  int32_t  seqId;
  int32_t  frag;              /* index of the query fragment in this call */
  int64_t  rangeStartPos;
  int64_t  rangeEndPos;
  int32_t  intersectionSize;
  int32_t  pad_;
} wfm_l1_candidate_t;

typedef struct {
  int32_t  window_length;          /* Parameters::windowLength (= segment length; fragments are exactly this long) */
  int32_t  sketch_size;            /* Parameters::sketchSize */
  int32_t  min_hits_cached;        /* cached_minimum_hits (computeMap.hpp:155-160) */
  int32_t  cached_segment_length;  /* cached_segment_length */
  int32_t  skip_self, skip_prefix, lower_triangular;
  int32_t  stage1_topANI_filter, stage2_full_scan;
  int32_t  n_seq;                  /* number of sequences known to the SequenceIdManager */
  const int32_t* ref_group;        /* [n_seq] idManager.getRefGroup(seqId) */
  const int32_t* min_hits_by_qsketch; /* [sketch_size+1] Stat::estimateMinimumHitsRelaxed(q, k, ANI) for the non-cached length */
  const int32_t* sketch_cutoffs;   /* [n_cutoffs] Stat::sketch_cutoffs table (computeMap.hpp:182-186) */
  int32_t  n_cutoffs;
  int32_t  pad_;
} wfm_l1_params_t;

/* qsketch: nfrag x s minmers (layout of wfm_sketch_fragments), qcount[f] of them valid.
 * q_active[f] == 0 skips a fragment (kmerComplexity below the threshold, computeMap.hpp:951).
 * Returns the number of candidates (may exceed cap; only cap are written) or a WFM_E_* code. */
int64_t wfm_map_l1(wfm_handle_t* h, const wfm_index_t* ix, const wfm_minmer_t* qsketch, const int32_t* qcount,
                   const int32_t* q_seq_id, const int32_t* q_len, const uint8_t* q_active, int64_t nfrag, int s,
                   const wfm_l1_params_t* prm, wfm_l1_candidate_t* out, int64_t cap);

/* L2 stage: doL2Mapping + computeL2MappedRegions + SlideMapper (computeMap.hpp:989-1061,
 * mappingCore.hpp:307-442, slidingMap.hpp:28-212) for a batch of L1 candidates.
 * One mapping = skch::MappingResult (base_types.hpp:154-165), 28 bytes. */
typedef struct {
  uint32_t refSeqId;
  uint32_t refStartPos;
  uint32_t queryStartPos;      /* 0: relative to the fragment; the caller adds fragmentIndex * windowLength (computeMap.hpp:124-128) */
  uint32_t blockLength;
  uint32_t n_merged;
  uint32_t conservedSketches;
  uint16_t nucIdentity;        /* x 1e4 */
  uint8_t  flags;              /* bit 0: reverse strand, bit 1: discard, bit 2: overlapped */
  uint8_t  kmerComplexity;     /* x 100 */
} wfm_mapping_t;

typedef struct {
  int32_t  window_length;
  int32_t  sketch_size;            /* Parameters::sketchSize = S; tables below are (S+1) x (S+1), index Q.sketchSize * (S+1) + shared */
  int32_t  stage1_topANI_filter;
  int32_t  pad_;
  const uint8_t*  keep_table;      /* identity test of computeMap.hpp:1018-1024 */
  const uint16_t* ident_table;     /* MappingResult::setNucIdentity(1 - j2md(shared / Q.sketchSize)) */
  const double*   cutoff_j;        /* [S+1] Jaccard cutoff of computeMap.hpp:1001-1004 per Q.sketchSize */
} wfm_l2_params_t;

/* q_kmer_complexity[f]: MappingResult::setKmerComplexity(Q.kmerComplexity) already scaled.
 * Mappings come out grouped by fragment, sorted by (refSeqId, refStartPos) within a fragment
 * (computeMap.hpp:920-921); out_frag[i] is the fragment of out[i].  Returns the number of
 * mappings (may exceed cap; only cap are written) or a WFM_E_* code. */
int64_t wfm_map_l2(wfm_handle_t* h, const wfm_index_t* ix, const wfm_minmer_t* qsketch, const int32_t* qcount,
                   const int32_t* q_len, const uint8_t* q_kmer_complexity, int64_t nfrag, int s,
                   const wfm_l1_candidate_t* cands, int64_t ncand, const wfm_l2_params_t* prm,
                   wfm_mapping_t* out, int32_t* out_frag, int64_t cap);

/* The fused per-batch map path: Map::mapSingleQueryFrag (computeMap.hpp:875-938) for nfrag
 * fragments of window_length bases each, fragment f = seq[frag_off[f] .. +window_length) belonging
 * to query sequence frag_seq_id[f].  Sketches, interval points and candidates never leave the
 * device.  Output as wfm_map_l2. */
typedef struct {
  int32_t  kmer_size;
  float    kmer_complexity_threshold;   /* Parameters::kmerComplexityThreshold (computeMap.hpp:951) */
  wfm_l1_params_t l1;
  wfm_l2_params_t l2;
} wfm_map_params_t;
int64_t wfm_map_fragments(wfm_handle_t* h, const wfm_index_t* ix, const char* seq, int64_t seq_len,
                          const int64_t* frag_off, const int32_t* frag_seq_id, int64_t nfrag,
                          const wfm_map_params_t* prm, wfm_mapping_t* out, int32_t* out_frag, int64_t cap);
/* The same, and the first step of the query's post-processing with it: mergeMappingsInRange[WithChains]
 * (mappingFilter.hpp:402-421, :593-612) begins by sorting a query's mappings by (target, strand, query position, target position);
 * the mappings are still on the device when L2 ends, so the batch is sorted there.  frag_first[f] = the first fragment of fragment f's
 * query (the caller adds (f - frag_first[f]) * window_length to queryStartPos: the key is made of that sum).  out_perm[i] = index into
 * out[] of the i-th mapping in (query, target, strand, query position, target position) order; out_perm[0] = 0xffffffff when two
 * mappings of a query share a key (std::sort leaves ties in an order of its own: the caller sorts as before) or the order could not
 * be made.  out / out_frag stay in fragment order, the order the reference's representative ids count in. */
int64_t wfm_map_fragments_ordered(wfm_handle_t* h, const wfm_index_t* ix, const char* seq, int64_t seq_len,
                                  const int64_t* frag_off, const int32_t* frag_seq_id, int64_t nfrag,
                                  const wfm_map_params_t* prm, wfm_mapping_t* out, int32_t* out_frag, int64_t cap,
                                  const int32_t* frag_first, uint32_t* out_perm);

#ifdef __cplusplus
}
#endif
#endif
