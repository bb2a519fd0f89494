/*
 * oracle/wfa2p.h -- TEST INFRASTRUCTURE ONLY (parity oracle / cpu_baseline "port").
 *
 * CPU restatement of the gap-affine 2-piece wavefront alignment that wfmash
 * obtains from the third-party module smarco/WFA2-lib (un-vendored submodule
 * `deps/WFA2-lib`, .gitmodules:1-3; pinned commit unrecoverable, the directory
 * is empty in /root/reference).  The algorithm follows the published WFA
 * (Marco-Sola 2021) and BiWFA (Marco-Sola 2023) papers and the WFA2-lib
 * conventions recalled in SURVEY.md Appendix A; parity is anchored on the
 * reference's own call sites:
 *   wflign.cpp:136-148   WFAlignerGapAffine2Pieces(0,x,o1,e1,o2,e2,Alignment,
 *                        MemoryUltralow).alignEnd2End(target,tlen,query,qlen)
 *   wflign.cpp:280-305   head patch  alignEndsFree(..., MemoryMed)
 *   wflign.cpp:368-397   tail patch  alignEndsFree(..., MemoryMed)
 *
 * PARITY UNPINNED at the CIGAR level: the reference holds no golden vector for
 * this boundary.  What IS pinned here: optimal score vs
 * an independent O(nm) DP (wfo_dp_score), CIGAR validity against the
 * sequences, and CIGAR-implied score == optimal score.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this file.  The product path never does.
 */
#ifndef ORACLE_WFA2P_H_
#define ORACLE_WFA2P_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t x;   /* mismatch           (parse_args.hpp:290  default 5)  */
  int32_t o1;  /* gap_opening1       (parse_args.hpp:291  default 8)  */
  int32_t e1;  /* gap_extension1     (parse_args.hpp:292  default 2)  */
  int32_t o2;  /* gap_opening2       (parse_args.hpp:293  default 24) */
  int32_t e2;  /* gap_extension2     (parse_args.hpp:294  default 1)  */
} wfo_penalties_t;

/* wavefront components (WFA2-lib affine2p_matrix_type order) */
enum { WFO_M = 0, WFO_I1 = 1, WFO_I2 = 2, WFO_D1 = 3, WFO_D2 = 4 };

typedef struct {
  uint64_t cells;        /* sum over computed (score,diagonal) cells, all 5 components counted once */
  uint64_t extend_bases; /* bases matched by extend */
  uint32_t bialign_calls;
  uint32_t base_calls;
  uint32_t max_depth;
} wfo_stats_t;

/* Independent O(plen*tlen) dynamic program; global alignment; a gap of
 * length L costs min(o1+L*e1, o2+L*e2); mismatch x; match 0.
 * (cost model: wflign_alignment.cpp:680-722). Returns the optimal score. */
int64_t wfo_dp_score(const char* pattern, int plen, const char* text, int tlen,
                     const wfo_penalties_t* pen);

/* Same DP with free ends: *_begin_free / *_end_free as in
 * WFAligner::alignEndsFree (see wfo_align_endsfree). */
int64_t wfo_dp_score_endsfree(const char* pattern, int plen, const char* text, int tlen,
                              const wfo_penalties_t* pen,
                              int pattern_begin_free, int pattern_end_free,
                              int text_begin_free, int text_end_free);

/* alignEnd2End with MemoryUltralow (BiWFA).  ops_out must hold plen+tlen+1
 * bytes; receives a NUL-terminated op string over {M,X,I,D} ('I' consumes
 * text, 'D' consumes pattern; wfmash passes pattern=target, text=query).
 * Returns 0 on success (WF_STATUS_ALG_COMPLETED, wflign.cpp:150). */
int wfo_align_end2end_biwfa(const char* pattern, int plen, const char* text, int tlen,
                            const wfo_penalties_t* pen,
                            char* ops_out, int* nops, int* score, wfo_stats_t* stats);

/* alignEnd2End, unidirectional WFA with full backtrace (MemoryHigh); used to
 * cross-check BiWFA scores and as the BiWFA base case. */
int wfo_align_end2end_uni(const char* pattern, int plen, const char* text, int tlen,
                          const wfo_penalties_t* pen,
                          char* ops_out, int* nops, int* score, wfo_stats_t* stats);

/* alignEndsFree (MemoryMed in wfmash; tie-breaks equal the full backtrace,
 * see wfa2p.c).  Output is the full-length op string: free prefixes/suffixes
 * appear as leading/trailing I/D runs (wflign.cpp:309,401). */
int wfo_align_endsfree(const char* pattern, int plen, int pattern_begin_free, int pattern_end_free,
                       const char* text, int tlen, int text_begin_free, int text_end_free,
                       const wfo_penalties_t* pen,
                       char* ops_out, int* nops, int* score, wfo_stats_t* stats);

/* Sub-problem entry used by the parity tests of the device base-case kernel:
 * unidirectional end2end alignment that begins in component comp_begin and
 * ends in component comp_end (BiWFA halves). */
int wfo_align_end2end_comp(const char* pattern, int plen, const char* text, int tlen,
                           const wfo_penalties_t* pen, int comp_begin, int comp_end,
                           char* ops_out, int* nops, int* score, wfo_stats_t* stats);

/* BiWFA breakpoint search only (parity tests of the device breakpoint kernel).
 * Returns 0 if a breakpoint was found, 1 if the end was reached at score 0
 * (caller falls back to the base case), <0 on error. */
typedef struct {
  int32_t score, score_forward, score_reverse;
  int32_t k_forward, k_reverse, offset_forward, offset_reverse;
  int32_t component;
} wfo_breakpoint_t;
int wfo_find_breakpoint(const char* pattern, int plen, const char* text, int tlen,
                        const wfo_penalties_t* pen, int comp_begin, int comp_end,
                        wfo_breakpoint_t* bp, wfo_stats_t* stats);
/* The same search under an upper bound `sub` of the score, as the PRODUCT runs it (not the reference): rows only hold the
 * diagonals from which the end diagonal is within reach, |k - (tlen - plen)| <= sub - s, and the overlap loop starts as
 * if a breakpoint of score sub + 1 were in hand.  The claim under test: for sub >= the unbounded breakpoint's score the
 * result is the unbounded one, field for field; below it nothing is found (a negative status). */
int wfo_find_breakpoint_bounded(const char* pattern, int plen, const char* text, int tlen,
                                const wfo_penalties_t* pen, int comp_begin, int comp_end, int sub,
                                wfo_breakpoint_t* bp, wfo_stats_t* stats);
/* The same search with its overlap loop cut every `tests_per_round` tests the way the product cuts it (phase 2 in rounds:
 * the breakpoint so far set aside, the next round seeded with its score); sub < 0: no bound.  *rounds = rounds taken. */
int wfo_find_breakpoint_rounds(const char* pattern, int plen, const char* text, int tlen,
                               const wfo_penalties_t* pen, int comp_begin, int comp_end, int sub, int tests_per_round,
                               wfo_breakpoint_t* bp, int* rounds, wfo_stats_t* stats);

/* Score implied by an op string under the reference's cost model
 * (wflign_alignment.cpp:680-722: a gap run of length L costs
 * o1+e1+min(e1*(L-1), o2+e2*(L-1)) ... note this equals min(o1+L*e1,o2+o1... )
 * only for the default penalties; wfo_ops_score uses min(o1+L*e1,o2+L*e2),
 * the quantity WFA optimises). */
int64_t wfo_ops_score(const char* ops, int nops, const wfo_penalties_t* pen);

/* pafcheck-style validation: every M column equal, every X column different,
 * ops consume exactly plen / tlen.  Returns 0 if valid, else 1-based index of
 * the first offending op (or -1 for a length mismatch). */
int wfo_ops_check(const char* ops, int nops, const char* pattern, int plen,
                  const char* text, int tlen);

/* Batch driver for bench.py's cpu_baseline leg: aligns n problems with
 * nthreads OpenMP threads (one problem per thread, mirroring the reference's
 * Taskflow for_each, computeAlignments.hpp:391-435). Sequences are packed in
 * one buffer; offsets/lengths index into it.  ops for problem i are written at
 * ops_arena + ops_off[i] (capacity plen+tlen+1).  Returns #failed. */
int wfo_align_batch_biwfa(const char* seqs, const int64_t* p_off, const int32_t* p_len,
                          const int64_t* t_off, const int32_t* t_len, int n,
                          const wfo_penalties_t* pen, char* ops_arena, const int64_t* ops_off,
                          int32_t* nops, int32_t* scores, int nthreads, wfo_stats_t* stats_sum);

#ifdef __cplusplus
}
#endif
#endif
