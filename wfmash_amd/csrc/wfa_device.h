// wfa_device.h -- job descriptors shared by the HIP kernels and the host driver.
#ifndef WFM_WFA_DEVICE_H_
#define WFM_WFA_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wfm {

constexpr int WF_NULL = -(1 << 30);
constexpr int RING = 32;  // rows kept per component; must be >= score scope.  The default penalties (scope 26) and everything
constexpr int RMASK = RING - 1;  // built for them (register tiles, phase 2 from rows computed ahead) use this depth
constexpr int RING_BIG = 128;  // the depth for other penalties whose scope passes 32 (o2 + e2 up to 125): step kernel + base kernel only

// wavefront components (same numbering as oracle/wfa2p.h)
enum { C_M = 0, C_I1 = 1, C_I2 = 2, C_D1 = 3, C_D2 = 4 };
// RLE op codes: entry = (len << 2) | op ; 0 = empty slot
enum { OP_M = 0, OP_X = 1, OP_I = 2, OP_D = 3 };
// backtrace decision byte: bits 0-2 = M source component, bits 3-6 = "came by extension"
enum { BT_I1_EXT = 8, BT_I2_EXT = 16, BT_D1_EXT = 32, BT_D2_EXT = 64 };

constexpr int WFM_DEV_UNREACHABLE = -300;
constexpr int WFM_DEV_OVERFLOW = -2;  // base job exceeded its score budget (smax)
constexpr int WFM_DEV_BAND = -4;      // bialign job ran out of its diagonal band (BpJob::band)
constexpr int SUB_NONE = 1 << 29;     // "no upper bound of the score is known" (BpJob::sub, TileJob::sub, P2Job::sub)
constexpr int WFM_DEV_P2_NOTHING = -6; // phase 2 ended without improving on the breakpoint it was handed (BpJob / P2Job::best0): that one stands
constexpr int WFM_DEV_P2_MORE = -5;   // phase 2 did not end within the P2K rows computed ahead: the step kernel takes the job

struct DevPen { int x, o1, e1, o2, e2; };

struct BpJob {
  int64_t p_fwd, t_fwd, p_rev, t_rev;  // byte offsets of the sub-range starts in the sequence buffer
  int64_t ring_off;                    // int32 element offset of this job's ring
  int32_t pl, tl;
  int32_t comp_begin, comp_end;
  int32_t width;                       // row stride = round4(pl + tl + 9)
  int32_t koff;                        // column = k + koff (multiple of 4; normally pl + 4)
  int32_t resume_s;                    // -1: start at score 0; >= 0: the forward direction resumes at this score
  int32_t fmax0, rmax0;                // running max antidiagonals at the resume point
  int32_t band;                        // > 0: the ring only holds diagonals |k| <= band + 8 (a job that is expected to end at a low
                                       // score gets a narrow ring); a direction that would pass score `band` ends the job with
                                       // WFM_DEV_BAND and the host runs it again on a full ring.  resume_s == -3: the tile phase
                                       // already ran out of the band
  // resume_sr >= 0: the snapshot is the exact meeting point found by the tile kernels -- forward at resume_s,
  // reverse at resume_sr (= resume_s or resume_s - 1), phase 1 is over; -1: both directions resume at resume_s
  int32_t resume_sr;
  int32_t last_fwd;                    // with resume_sr >= 0: 1 if the forward step was the last one taken
  int32_t sub;                         // upper bound of the job's score (SUB_NONE: none): rows only hold |k - (tl - pl)| <= sub - s
  int32_t best0;                       // > 0: phase 2 resumes with a breakpoint of this score in hand (found by earlier rounds of rows
                                       // computed ahead); only a better one is reported, else WFM_DEV_P2_NOTHING
  int32_t packed;                      // bit 0: both sequences are pure upper-case ACGT -- the tile kernel may read the 2-bit mirror (wfa_tile2.hip);
                                       // bit 1: near-identical sequences (score known to be under a sixteenth of the length): long runs go to the wave's tail at once
};

// ---- time-tiled phase 1 (wfa_tile_kernel) ----
// A block advances every active job by T scores.  A tile owns `core` diagonals of one
// direction of one job, loads core + 2T columns of the snapshot (scope M rows, e1 rows of
// I1/D1, e2 rows of I2/D2) into LDS, runs T steps there (trapezoid: the computed region
// shrinks by one column per side per step) and writes the last `scope` rows of all five
// components of its core back.  Row ranges are closed-form: score s covers
// diagonals [max(-pl,-s), min(tl,s)].
struct TileJob {
  int64_t p_fwd, t_fwd, p_rev, t_rev;
  int64_t ring_in, ring_out;           // int32 element offsets of the two snapshot rings
  int32_t pl, tl;
  int32_t comp_begin, comp_end;
  int32_t width, koff;
  int32_t s0;                          // snapshot score of ring_in
  int32_t active;                      // 0: the meeting point lies in the block after s0 (or the job is over): tiles exit
  int32_t fmax, rmax;                  // running maximum antidiagonals of the two directions up to s0
  int32_t nblocks;                     // tile blocks executed so far (incl. the one that found the meeting point)
  int32_t mode;                        // 0: full blocks; 1: next block stops exactly at the meeting point; 2: stopped there; 6: the block BEFORE s0 runs again (ring_prev, below);
                                       // 5: the block after s0 runs again with per-score maxima (it ran with one maximum for the whole block
                                       // and the wavefronts met inside it: fine_s below)
  int32_t tf, tr;                      // mode >= 1: steps of the forward / reverse direction inside the block after s0
  int32_t last_fwd;                    // mode >= 1: 1 if the forward check ended phase 1 (reverse is one step behind)
  int32_t packed;                      // as BpJob::packed (the job's tiles run wfa_tile2_kernel)
  // mode 4 (phase-2 rows, see P2Job): forward starts at score tf, reverse at tr (s0 is not used), every row goes to the
  // job's P2 rows instead of an output snapshot
  int64_t p2_off;
  int32_t w2, koff2;
  int32_t sub;                         // as BpJob::sub
  // Round 6: the per-score maximum antidiagonals (a six-step DPP reduction per wave and score, a seventh of the step's vector issue) are only
  // needed in the block in which the wavefronts meet.  A block that ends below score fine_s keeps ONE maximum per direction (monotone running
  // maxima: fm + rm >= A at the block's end <=> the directions met somewhere inside it); the advance kernel runs a block that met with a single
  // maximum again with per-score maxima (mode 5) -- a child's score is known, so its last blocks are fine from the start and nothing runs three times.
  int32_t fine_s;
  // Round 6: a block used to write the gap components' rows of its last 26 scores into its output snapshot -- 104 values per diagonal, 60 % of the
  // tile kernel's HBM traffic -- for ONE reader: the run up to the meeting point of the NEXT block, when that run is shorter than 26 scores and
  // its own output (which phase 2 reads 26 rows deep in all five components) needs rows from before its start.  With a third ring the input
  // of the block before (ring_prev) survives one block longer: blocks write the two rows (I1 / D1) and one row (I2 / D2) the next block loads,
  // and when a run up to the meeting point is short, the block before it runs once more WITH its gap rows (mode 6: from ring_prev into ring_out)
  // and the run starts from that.  ring_prev < 0: no third ring (the chunk did not fit one) -- every block writes the 26 rows as before.
  int64_t ring_prev;
  int32_t prev_ok;                     // ring_prev holds the snapshot of score s0 - T (false before the phase's second block)
  int32_t reran;                       // mode-6 runs so far (the host's cell count)
};

// ---- phase 2 (overlap detection) without a step-by-step kernel ----
// wavefront_bialign_find_breakpoint's second loop alternates "test the newest row of one direction against the last
// `scope` rows of the other" and "advance the other direction by one row" until no better breakpoint is possible
// -- about 2 * scope rows past the meeting point.  The rows do not depend on the tests, so they are computed ahead:
// the tile kernel runs P2K more scores of both directions from the exact snapshot and keeps EVERY row (five components,
// [dir][comp][P2K][w2]); wfa_p2_blockmax_kernel takes the maxima of every row per component and per block of 64
// diagonals (a pair of rows can only meet where both are far along: the block maxima prune by POSITION, which the row
// maxima cannot -- a direction that has already crossed most of the text would otherwise let every diagonal of the other
// through); wfa_p2_overlap_kernel then walks the reference's loop, one workgroup per job, with nothing but
// the tests left in it (doing all tests of a job side by side was tried: without the best breakpoint so far to prune
// with, the tests after the first hit cost more than the whole sequential walk).  A job whose loop has not ended after
// 2 * P2K tests (WFM_DEV_P2_MORE) is finished by wfa_bp_kernel from the same snapshot.
constexpr int P2K = 32;            // (48 until jobs could take further rounds: most walks end within 2 x 26 tests, and the rows of the
                                    // others are computed when they are asked for -- C3 109 -> 106 ms per step, C1 substitute 5.5 -> 5.4 s)
constexpr int P2ROWS = 26 + P2K;   // row maxima per direction: the snapshot's rows sd-25 .. sd, then sd+1 .. sd+P2K
constexpr int P2TESTS = 2 * P2K;
constexpr int P2ENT = RING * 5;    // (row of the other direction, component) slots of a test; scope <= RING rows are used
struct P2Job {
  int64_t ring_in;                     // the exact snapshot (rows <= sf / sr), int32 element offset of the job's ring
  int64_t p2_off;                      // int32 element offset of the job's P2 rows
  int32_t width, koff;                 // ring geometry
  int32_t w2, koff2;                   // P2 geometry: column = k + koff2
  int32_t pl, tl;
  int32_t sf, sr, last_fwd;            // state at the meeting point
  int32_t sub, best0;                  // as BpJob::sub, BpJob::best0
  int32_t nblk;                        // 64-diagonal blocks of a row: block of diagonal k = (k + koff2) >> 6
  int64_t bm_off;                      // int32 element offset of the job's block maxima [dir][P2ROWS][comp][nblk]
};
struct TileTask {
  int32_t job, dir;
  int32_t core_lo, core_hi;            // in memory: (tile index, tile width); the kernels turn it into the inclusive
                                       // diagonal range owned by the tile for the block at hand
};

struct BpResult {
  int32_t status;  // 0 breakpoint found; 1 end reached at score 0; <0 error
  int32_t score, score_fwd, score_rev, k_fwd, off_fwd, comp;
  int32_t steps;
  uint64_t cells;
  int32_t steps_p1;            // steps spent before the antidiagonals met
  uint32_t ticks_p1, ticks_p2; // wall_clock64 ticks (100 MHz) per phase
  int32_t pad_;
  uint32_t ticks_list, ticks_pick;  // wfa_p2_overlap_kernel: the block-listing and the pick stages (diagnostics)
  uint32_t work_items, pad2_;       // blocks listed over all rounds
};

struct BaseJob {
  int64_t p_off, t_off;  // byte offsets of the (forward) sub-range starts
  int64_t pre_off;       // int32 element offset: pre[(smax+1)][width]
  int64_t bt_off;        // byte offset: bt[(smax+1)][width]
  int64_t ring_off;      // int32 element offset: ring[5][RING][width]
  int64_t rle_end;       // exclusive end of this job's slot in the RLE buffer
  int32_t pl, tl;
  int32_t comp_begin, comp_end;
  int32_t endsfree, pbf, pef, tbf, tef;
  int32_t smax, kmin, width;
  int32_t type;          // 0 WFA; 1 all-D (tl == 0); 2 all-I (pl == 0)
  int32_t pad_;
};

struct BaseResult {
  int32_t status;  // 0 ok; WFM_DEV_OVERFLOW; <0 error
  int32_t score;
  int32_t nruns;
  int32_t pad_;
  uint64_t cells;
};

// ---- base jobs whose rows are wider than one workgroup's registers hold (wfa_base2t_kernel, round 6) ----
// A patch that overflowed its budget of 1020 ran on r32::wfa_base_kernel<1024> until round 6: one workgroup per job, 6 us per score step over rows of
// 3 k diagonals from a global-memory ring, a dozen jobs on a device of 256 CUs for 9 ms at the end of every batch.  Here the columns of such a job are
// cut into tiles as the tile phase of BiWFA cuts its rows: a launch is one BLOCK of T scores, every tile a workgroup that holds core + 2 T diagonals in
// the register kernel's delay lines (wfa_base2_kernel's step, its decisions, its rows of pre / bt -- written for the core only), T columns of halo on
// either side that it computes for itself and that go wrong one column per step from the outside, a snapshot of the last 26 rows between two blocks.
// A tiny kernel between two launches replays the end test over the tiles of a job (the first score at which a cell ends, the smallest such diagonal),
// and one wave per job walks back through pre / bt at the end (base2_walk: the register kernel's own walk).
struct Base2TJob {
  BaseJob b;                   // as for wfa_base2_kernel; b.ring_off: two snapshots of B2T_ROWS rows x width behind each other
  int64_t snap_in, snap_out;   // int32 element offsets of the snapshot the next block loads / writes
  int32_t ntiles, core;        // tiles of `core` diagonals from b.kmin on
  int32_t task0;               // the job's first entry in the task list (its tiles follow each other)
  int32_t s0;                  // score the next block starts from (0: row 0 is still to be made)
  int32_t done;                // 0 running, 1 an end was found (end_s / end_k / end_off), 2 the budget is spent
  int32_t end_s, end_k, end_off;
};
struct Base2TTask { int32_t job, tile; };
constexpr int B2T_ROWS = 32;   // rows of a snapshot: M of the last 26 scores, I1 / D1 of the last two, I2 / D2 of the last
constexpr int B2T_THREADS = 512;
void launch_base2t_block(const uint32_t* pk, int32_t* a32, uint8_t* a8, const Base2TJob* jobs, const Base2TTask* tasks, unsigned long long* tile_key,
                         int32_t* tile_off, int ntasks, int T, hipStream_t st);
void launch_base2t_advance(Base2TJob* jobs, const unsigned long long* tile_key, const int32_t* tile_off, int njobs, int T, int32_t* active_slot, hipStream_t st);
void launch_base2t_finish(const int32_t* a32, const uint8_t* a8, uint32_t* rle, const Base2TJob* jobs, BaseResult* res, int njobs, hipStream_t st);

// ---- an upper bound of a root's score before its wavefronts are computed (wfa_bound_kernel) ----
struct BoundJob { int64_t p_off, t_off; int32_t pl, tl; };
void launch_bound(const uint8_t* seq, const BoundJob* jobs, int32_t* out, int njobs, DevPen pen, hipStream_t st);

// reversed copies of the sequences of a BiWFA problem, made on the device (wfm_upload_sequences)
struct SeqRev { int64_t p_fwd, p_rev, t_fwd, t_rev; int32_t plen, tlen; };
void launch_reverse(uint8_t* seq, const SeqRev* jobs, int njobs, int pad, hipStream_t st);
// the 2-bit mirror of the sequence buffer (word i = bytes 16 i .. 16 i + 15) and, per BiWFA problem, "pure ACGT" (flag stays nonzero)
void launch_seq_pack(const uint8_t* seq, uint32_t* pk, int64_t nwords, int64_t nbytes, const SeqRev* jobs, int njobs, int32_t* flag, hipStream_t st);
constexpr int64_t PK_PAD_WORDS = 2048 + 64;  // words of padding behind the mirror: a tile stages a whole window from any origin inside
// the tile kernel on packed sequences (wfa_tile2.hip): same contract as launch_tile_reg / launch_tile_p2
void launch_tile2(const uint32_t* pk, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int32_t* mak, int ntasks, int threads, int T, int variants, hipStream_t st);
bool tile2_coarse_maxima();  // wfa_tile2_kernel keeps one maximum per block below TileJob::fine_s (the FAST form; WFM_TILE_COARSE=0: per score always)
void launch_tile2_p2(const uint32_t* pk, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int ntasks, int threads, int32_t* p2, hipStream_t st);
int selftest_dpp(int* host_out128, hipStream_t st);
// leaves and patches on registers and packed sequences (default penalties; rows of at most 2 * threads diagonals): launch_base's contract
void launch_base2(const uint32_t* pk, int32_t* a32, uint8_t* a8, uint32_t* rle, const BaseJob* jobs, BaseResult* res, int njobs, int threads, hipStream_t st);
void launch_bp(const uint8_t* seq, int32_t* ring, const BpJob* jobs, BpResult* res, int njobs, int threads,
               DevPen pen, int scope, int ring_rows, hipStream_t st);
void launch_tile_init(const uint8_t* seq, int32_t* ring, const TileJob* jobs, int32_t* mak0, int njobs, int ring_rows, hipStream_t st);
void launch_tile(const uint8_t* seq, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int32_t* mak, int ntasks,
                 int threads, int T, int Wt, size_t lds_bytes, DevPen pen, int scope, int ring_rows, hipStream_t st);
void launch_tile_advance(TileJob* jobs, int32_t* mak, int njobs, int T, DevPen pen, int exact, int coarse, int launched, hipStream_t st);
void launch_tile_reg(const uint8_t* seq, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int32_t* mak, int ntasks,
                     int threads, int T, int C, bool cut, hipStream_t st);  // cut: some job of the launch carries a score bound
// phase-2 rows of the jobs in mode 4 (T = P2K scores, two diagonals per thread), their per-row maxima into p2max
void launch_tile_p2(const uint8_t* seq, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int ntasks, int threads,
                    int32_t* p2, bool cut, hipStream_t st);
// The state after 2 * P2K tests as a snapshot: the last RING rows of both directions, out of the P2 rows into the job's ring
// (jobs whose walk ran out of rows go another round from there)
void launch_p2_to_ring(int32_t* ring, const int32_t* p2, const P2Job* jobs, int njobs, hipStream_t st);
void launch_p2_blockmax(const int32_t* ring, const int32_t* p2, const P2Job* jobs, int32_t* bmax, int32_t* p2max, int njobs, hipStream_t st);
void launch_p2_overlap(const int32_t* ring, const int32_t* p2, const P2Job* jobs, const int32_t* p2max, const int32_t* bmax, int32_t* pbmax,
                       BpResult* res, int njobs, int threads, int max_nblk, DevPen pen, int scope, hipStream_t st);
void launch_base(const uint8_t* seq, int32_t* a32, uint8_t* a8, uint32_t* rle, const BaseJob* jobs, BaseResult* res,
                 int njobs, DevPen pen, bool wide, int ring_rows, hipStream_t st);  // wide: 1024 threads per job instead of 256
void launch_compact(const uint32_t* rle, const int64_t* off, const int64_t* cap, uint32_t* out, unsigned long long* total,
                    int64_t* out_start, int32_t* out_count, int nprob, hipStream_t st);

}  // namespace wfm
#endif
