/*
 * include/wfmash_host.h -- file-level seam of the align phase (C ABI).
 *
 * wfmh_align_paf replaces align::Aligner::compute()
 * (src/align/include/computeAlignments.hpp:185,318-455): it reads a mapping PAF
 * (the hand-off file between the reference's map and align phases, `-i file.paf`,
 * src/interface/parse_args.hpp:800-804), fetches the sequence windows, runs the
 * wflign pipeline (BiWFA + head/tail patches + swizzle) on the GPU and writes the
 * aligned PAF.  Defaults equal the reference's (parse_args.hpp:290-294,566-620).
 */
#ifndef WFMASH_HOST_H_
#define WFMASH_HOST_H_

#include <stdint.h>
#include "wfmash_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t  mismatch, gap_open1, gap_ext1, gap_open2, gap_ext2;  /* --wfa-params, default 5,8,2,24,1 */
  float    min_identity;            /* 0 */
  uint64_t min_alignment_length;    /* 32 */
  float    min_block_identity;      /* 0.1 */
  uint64_t target_padding;          /* min(w,5000) = 1000 */
  uint64_t query_padding;           /* min(w,5000) = 1000 */
  uint64_t wflign_max_len_minor;    /* 128 * w = 128000 */
  int32_t  disable_chain_patching;  /* 0 */
  int32_t  sam_format;              /* -a: SAM instead of PAF (parse_args.hpp:128) */
  int32_t  emit_md_tag;             /* -d: MD:Z tag in SAM records (parse_args.hpp:129) */
  int32_t  no_seq_in_sam;           /* 0 */
  int32_t  threads;                 /* -t: host threads for fetching sequences and for the CIGAR / PAF work of a
                                       batch (the reference runs one Taskflow worker per record); 0 = all cores */
  int32_t  pad_;
} wfmh_align_params_t;

typedef struct {
  uint64_t records;      /* "total aligned records" */
  uint64_t aligned_bp;   /* "total aligned bp" (sum of query spans; computeAlignments.hpp:451-454) */
  uint64_t written;      /* PAF lines written */
  uint64_t skipped;      /* invalid rows */
  uint64_t cells;        /* wavefront cells computed on the GPU */
  double   ms_gpu;       /* time during which an align kernel was running (union over the streams and over the handles the
                            driver's workers use; the busiest device of a multi-GPU run) */
  double   ms_total;
  /* host stages, summed over the batches (batches of different workers overlap: the sums may exceed ms_total) */
  double   ms_rows;      /* parsing the mapping rows */
  double   ms_fetch;     /* fetching, normalising and strand-adjusting the sequence windows */
  double   ms_wflign;    /* the wflign pipeline of a batch: device calls (upload, alignment, patches) and the work on runs between them */
  double   ms_text;      /* collecting the records' text */
  uint64_t batches;
  /* the tile kernels' share (wfa_tile2_kernel / wfa_tile_reg_kernel, the dominant kernels of the path): cells the result needs
   * (the block a job computes twice counted once), launches, and the sum of the launches' durations -- launches of several
   * workers overlap, so the sum is an upper bound of the time the kernel had the device to itself */
  uint64_t cells_tile;
  uint64_t tile_launches;
  double   ms_tile;
  double   ms_tags;      /* WFM_RECORD_TAGS (a diagnostic channel of the parity tests and bench.py): writing the records' tags, summed over the batches */
} wfmh_align_summary_t;

void wfmh_align_default_params(wfmh_align_params_t* p);

/* query_fasta may be NULL (= target_fasta, all-vs-all).  Returns 0 or WFM_E_*;
 * messages go to stderr and wfm_last_error(h). */
int wfmh_align_paf(wfm_handle_t* h, const char* target_fasta, const char* query_fasta,
                   const char* mapping_paf, const char* out_paf,
                   const wfmh_align_params_t* params, wfmh_align_summary_t* summary);

/* The same over n GPUs of one node (handles[i] from wfm_create(device_i)): batches of mapping records go to whichever
 * device is free (records are independent, computeAlignments.hpp:398-435; the split the reference's cluster script
 * makes ahead of time, scripts/split_approx_mappings_in_chunks.py:19-27,47, is made at run time), the output is
 * written in input order and does not depend on n.  Errors are reported on handles[0]. */
int wfmh_align_paf_multi(wfm_handle_t* const* handles, int n, const char* target_fasta, const char* query_fasta,
                         const char* mapping_paf, const char* out_paf,
                         const wfmh_align_params_t* params, wfmh_align_summary_t* summary);

/* Test hooks: the pure host-side CIGAR functions (erode / merge / swizzle / PAF writer /
 * parseMashmapRow) behind one string interface so the CPU test-suite can check them
 * without a GPU.  fn in {erode, merge, compress, swap_start, swap_end, head_erosion,
 * tail_erosion, paf, parse_row}.  The result is malloc'd; release with wfmh_free. */
char* wfmh_test_cigar(const char* fn, const char* a, const char* b, const char* query, const char* target,
                      long long i0, long long i1);
void  wfmh_free(char* p);
/* host winnowing stage of wfm_add_minmers on caller-supplied canonical k-mer hashes (CPU tests) */
int64_t wfmh_test_winnow(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                         const uint64_t* hash, const int8_t* strand, wfm_minmer_t* out, int64_t cap);

/* the same through speculative chunks of chunk_len k-mers (how wfm_add_minmers_multi spreads one
 * long sequence over its workers); *replays = chunks whose speculation failed and were replayed
 * from the previous chunk's state, -1 = the sequence fell back to one stream */
int64_t wfmh_test_winnow_chunked(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                 const uint64_t* hash, const int8_t* strand, int64_t chunk_len,
                                 wfm_minmer_t* out, int64_t cap, int* replays);

/* the same on the thinned stream wfm_add_minmers_multi feeds its workers (wfm_prefilter_kmers,
 * wfmash_hip.h): the selection is restated on the host from its definition, with the hash threshold set to
 * let c_factor * s of a window's w-k+1 k-mers through, and only the kept k-mers are winnowed.  kept_pos (optional)
 * receives the kept k-mer starts, *n_kept their number. */
int64_t wfmh_test_winnow_thinned(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id,
                                 const uint64_t* hash, const int8_t* strand, double c_factor, int64_t chunk_len,
                                 wfm_minmer_t* out, int64_t cap, uint32_t* kept_pos, int64_t cap_kept,
                                 int64_t* n_kept, int* replays);

/* the closing sort of a sequence's records by (wpos, wpos_end) (commonFunc.hpp:696): threads == 1 is std::sort,
 * threads > 1 the same introsort with its halves on several threads -- same result, ties included */
void wfmh_test_sort_records(wfm_minmer_t* recs, int64_t n, int threads);
/* CPU test hooks of the 2-bit packed extension (wfmash_amd/csrc/wfa_pack.h): the run of agreeing bases from (v, h) of two sequences that begin at
 * byte start_p / start_t of their buffers, computed on the packed mirror the way the tile kernel stages it; whether a buffer is pure upper-case ACGT */
int wfmh_test_packed_lce(const uint8_t* buf_p, int64_t n_p, int64_t start_p, const uint8_t* buf_t, int64_t n_t, int64_t start_t, int v, int h, int maxn);
int wfmh_test_is_acgt(const uint8_t* seq, int64_t n);

/* The winnowing kernel's control flow and capacities (wfmash_amd/csrc/map_winnow_core.h: one speculative chunk of the
 * thinned stream per wave, boundary states compared, interval starts resolved) run on the host over plain arrays, then
 * the closing steps.  Returns the number of records, or -1 when the device would hand the sequence back to the host's
 * winnower (*why: the wn::F_* bits of map_winnow_core.h; bit 31: an N among the first k-mers the reference does not notice).
 * chunk_len < 0: chunks of -chunk_len k-mers, and every second speculation counts as failed (the replay from the
 * predecessor's state). */
int64_t wfmh_test_winnow_model(const char* seq, int64_t len, int k, int w, int s, int32_t seq_id, const uint64_t* hash,
                               const int8_t* strand, double c_factor, int64_t chunk_len, wfm_minmer_t* out, int64_t cap,
                               uint32_t* why);

/* The arrangement wfm_finish_records computes on the device -- the partitions of std::sort's introsort loop as lists,
 * swaps and a cut, then a stable sort (wfmash_amd/csrc/map_finish.hip) -- computed on the host, in place; key = (wpos, wpos_end). */
void wfmh_test_sortlike_model(wfm_minmer_t* recs, int64_t n);

/* The closing steps of addMinmers on raw interval records by the host path (cut, strand signs, std::sort, std::unique):
 * what wfm_finish_records (wfmash_hip.h) is held against. */
int64_t wfmh_test_finish_records(const wfm_minmer_t* raw, int64_t n, int w, wfm_minmer_t* out, int64_t cap);

/* ---- map phase (skch::Map, src/map/include/computeMap.hpp) ---- */

/* skch::Parameters as set up by parse_args.hpp; wfmh_map_default_params fills the defaults
 * (file:line of each default in wfmash_amd/host/map_types.hpp). */
typedef struct {
  int32_t  kmer_size;                 /* -k 15 */
  int64_t  window_length;             /* -w / segment length, 1000 */
  int64_t  block_length;              /* -l 0 */
  int64_t  chain_gap;                 /* -c 2000 */
  uint64_t max_mapping_length;        /* -P 50000 */
  float    percentage_identity;       /* -p as a fraction; 0.70 unless estimated */
  int32_t  sketch_size;               /* -s; 0 = derive from identity (parse_args.hpp:642-644) */
  int32_t  filter_mode;               /* 1 map (default), 2 one-to-one, 3 none */
  uint32_t num_mappings_for_segment;  /* -n; UINT32_MAX = inf (default) */
  uint32_t num_mappings_for_scaffold; /* 1 */
  int32_t  drop_rand;                 /* 0 */
  int32_t  split;                     /* 1 */
  int32_t  merge_mappings;            /* 1 */
  int32_t  skip_self;                 /* 1 */
  int32_t  skip_prefix;               /* 1 */
  int32_t  lower_triangular;          /* 0 */
  char     prefix_delim;              /* '#' */
  int32_t  filter_length_mismatches;  /* 1 */
  uint64_t sparsity_hash_threshold;   /* UINT64_MAX */
  double   overlap_threshold;         /* 0.95 */
  double   scaffold_overlap_threshold;/* 0.5 */
  int64_t  scaffold_max_deviation;    /* 100000 */
  int64_t  scaffold_gap;              /* 100000 */
  int64_t  scaffold_min_length;       /* 10000 */
  int32_t  legacy_output;             /* 0 */
  int32_t  minimum_hits;              /* 3 */
  double   max_kmer_freq;             /* 0.0002 */
  int64_t  index_by_size;             /* -b; INT64_MAX */
  float    kmer_complexity_threshold; /* 0 */
  int32_t  stage1_topani_filter;      /* 1 */
  int32_t  stage2_full_scan;          /* 1 */
  float    ani_diff, ani_diff_conf;   /* 0.0, 0.999 */
  double   hg_numerator;              /* 1.0 */
  int32_t  threads;                   /* 1 */
  int32_t  auto_pct_identity;         /* 1: estimate the identity from the data, -p aniN[+-adj] (parse_args.hpp:341-395);
                                         0: use percentage_identity as given */
  int32_t  ani_percentile;            /* 50 */
  float    ani_adjustment;            /* -2.0 (percent) */
  /* sequence selection (parse_args.hpp:112-115); NULL = unset */
  const char* target_prefix;          /* -T: only targets whose name starts with this */
  const char* target_list;            /* -R: file with target names */
  const char* query_prefix;           /* -Q: comma-separated name prefixes */
  const char* query_list;             /* -A: file with query names (also how ranks shard the queries) */
  /* on-disk index (parse_args.hpp:745-758; format: wfmash_amd/host/index_file.hpp) */
  const char* index_file;             /* NULL = none */
  int32_t  write_index;               /* 1: -W, build the index of every target subset, write it and stop;
                                         0 with index_file set: -I, read the index instead of building it */
  int32_t  pad_;
} wfmh_map_params_t;

void wfmh_map_default_params(wfmh_map_params_t* p);

typedef struct {
  uint64_t targets, queries, subsets;
  uint64_t target_bp, query_bp;
  uint64_t index_windows;   /* minmer intervals indexed (all subsets) */
  uint64_t fragments;       /* query fragments mapped (all subsets) */
  uint64_t l2_mappings;     /* MappingResults produced by the GPU stages */
  uint64_t written;         /* mapping PAF records written */
  float    percentage_identity;  /* the threshold used (estimated when auto_pct_identity) */
  int32_t  sketch_size;          /* the sketch size used */
  double   ms_index, ms_map, ms_filter, ms_total;
  double   ms_replicate;    /* copying the finished index to the other GPUs (wfmh_map_multi) */
  double   ms_identity;     /* the identity estimate (-p aniN: reading every sequence once + one MinHash per sequence on the device) */
  double   ms_wall;         /* the whole call: identity estimate, reading the sequences, index, mapping, post-processing, output */
} wfmh_map_summary_t;

/* The map phase on files: replaces skch::Map's constructor + mapQuery()
 * (src/map/include/computeMap.hpp:147-230, :329-872).  Reads the FASTA files (query_fasta NULL or
 * equal to target_fasta = all-vs-all), indexes the targets on the GPU, maps every query fragment
 * (sketch, L1, L2 on the GPU), post-processes per query on the host and writes the approximate
 * mapping PAF (the -m / -i hand-off file, parse_args.hpp:781-811).  Returns 0 or WFM_E_*. */
int wfmh_map(wfm_handle_t* h, const char* target_fasta, const char* query_fasta, const char* out_paf,
             const wfmh_map_params_t* params, wfmh_map_summary_t* summary);

/* The same over n GPUs of one node: the index of a target subset is built once (handles[0]) and copied to the other
 * devices (wfm_index_replicate), batches of whole query sequences go to whichever device is free (one task per query
 * in the reference, computeMap.hpp:527-688), records are written in query order as with one GPU. */
int wfmh_map_multi(wfm_handle_t* const* handles, int n, const char* target_fasta, const char* query_fasta, const char* out_paf,
                   const wfmh_map_params_t* params, wfmh_map_summary_t* summary);

/* Test hook for the host-side post-processing of one query's mappings (CPU tests): runs
 * mappingBoundarySanityCheck + Map::filterSubsetMappings + reportReadMappings
 * (stage "subset"), or filterByGroup on the reference axis as the one-to-one pass does
 * (stage "onetoone"), on caller-supplied MappingResults and returns the mapping PAF text
 * (malloc'd; wfmh_free).  fasta: the file whose .fai (or the FASTA itself) defines the sequences. */
char* wfmh_test_filter(const char* stage, const wfm_mapping_t* maps, int64_t n, const char* fasta,
                       const char* query_name, const wfmh_map_params_t* prm);

/* Test hook for the FASTA reader that stands in for faigz/htslib (src/common/faigz.h:221-505;
 * random access through .fai, and .gzi for BGZF).  name == NULL: "indexed|in-memory" + one
 * "name\tlength" line per sequence in file order.  Otherwise bases [start, end_inclusive] of `name`
 * (faidx_reader_fetch_seq's convention); whole != 0 loads the sequence first, as the map driver does.
 * malloc'd, wfmh_free; "ERROR: ..." on failure. */
char* wfmh_test_fasta(const char* path, const char* name, int64_t start, int64_t end_inclusive, int whole);

/* The sequences a map call loaded stay in memory for the call that follows (the align phase fetches its windows from the
 * same files; the reference keeps its faidx readers open for the whole run, src/common/faigz.h:221-505).  They are let go
 * when the next map call hands in its own, or here.  (WFM_FASTA_KEEP=0: nothing is kept; WFM_FASTA_KEEP_GB: the most that is, 32.) */
void wfmh_release_sequences(void);

/* Test hook: the whole sequence `name` through the store the process shares per path (open_shared), which is then kept as a
 * map call keeps its stores: a second call on a file rewritten in between must see the new bytes.  malloc'd, wfmh_free. */
char* wfmh_test_fasta_shared(const char* path, const char* name);

/* Test hook for the on-disk index format (wfmash_amd/host/index_file.hpp; no GPU needed).  op "ids": the id
 * section alone (SequenceIdManager::exportIdMapping, sequenceIds.hpp:101-115) of fasta's sequences into out_path;
 * op "rewrite": every sub-index of in_path is read and written again into out_path.  0, or -1 (message on stderr). */
int wfmh_test_index_file(const char* op, const char* fasta, char prefix_delim, const char* in_path, const char* out_path);

#ifdef __cplusplus
}
#endif
#endif
