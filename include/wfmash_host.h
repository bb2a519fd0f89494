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
} wfmh_align_params_t;

typedef struct {
  uint64_t records;      /* "total aligned records" */
  uint64_t aligned_bp;   /* "total aligned bp" (sum of query spans; computeAlignments.hpp:451-454) */
  uint64_t written;      /* PAF lines written */
  uint64_t skipped;      /* invalid rows */
  uint64_t cells;        /* wavefront cells computed on the GPU */
  double   ms_gpu, ms_total;
} wfmh_align_summary_t;

void wfmh_align_default_params(wfmh_align_params_t* p);

/* query_fasta may be NULL (= target_fasta, all-vs-all).  Returns 0 or WFM_E_*;
 * messages go to stderr and wfm_last_error(h). */
int wfmh_align_paf(wfm_handle_t* h, const char* target_fasta, const char* query_fasta,
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

#ifdef __cplusplus
}
#endif
#endif
