// wfa_host.hip -- host driver + C ABI of the align path (see include/wfmash_hip.h).
//
// Restates the control flow WFA2-lib runs on the CPU for
// WFAlignerGapAffine2Pieces::alignEnd2End(MemoryUltralow) (wflign.cpp:136-148):
// recursive BiWFA -- find breakpoint, split, recurse; sub-problems whose
// remaining score is <= 250 (or trivially empty) go to the unidirectional base
// aligner -- but breadth-first: every recursion level of every problem of the
// batch is ONE launch of wfa_bp_kernel plus ONE launch of wfa_base_kernel.
// Ends-free patches (wflign.cpp:280-305,368-397) are base jobs directly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "wfa_device.h"
#include "dev_cache.h"

#ifdef WFM_PROFILE_SECTIONS
namespace wfm { void read_sections(long long* out); }
#endif
namespace wfm { void p2_counters(unsigned long long* out); }
namespace {

using namespace wfm;

constexpr int BIALIGN_FALLBACK_MIN_SCORE = 250;   // WFA2-lib WF_BIALIGN_FALLBACK_MIN_SCORE
constexpr int BIALIGN_FALLBACK_MIN_LENGTH = 100;  // WFA2-lib WF_BIALIGN_FALLBACK_MIN_LENGTH
constexpr int SEQ_PAD = 64;  // extension reads up to 40 bytes past a sub-range end

#define HIPCHK(h, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return WFM_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  // The blocks come from, and go back to, the per-device block cache (dev_cache.h), never straight to the driver: memory a
  // process has freed is wiped before it is handed out again, and an allocation that lands on it waits for that -- 40 ms per GB
  // as a rule, 1.3 - 1.8 s at worst (profiles/r4_map_host.md).  An arena that is outgrown waits in the cache (its size class
  // serves the next handle, or the next batch's sequences); at most as much again as the final arena is held that way.
  int ensure(size_t n) {
    if (n <= cap) return 0;
    const size_t old_cap = cap;
    if (p) wfm_dfree(p);  // (waits for the device, as hipFree did)
    p = nullptr; cap = 0;
    // (an arena that has to grow doubles at least: every regrowth is a fresh allocation, 30 - 70 ms per GB on this driver, and a
    // divergent batch -- C1 -- used to walk its ring arena up in five steps of 3.5 .. 12 GB)
    // ... but never by more than 8 GB beyond what is asked for: the arenas live under per-handle budgets of up to 32 GB, and a ring arena near
    // its budget that doubled held 64 GB + the outgrown 32 in the cache, outside every budget's accounting)
    size_t want = std::max(n + n / 8 + 64, std::min(old_cap * 2, n + ((size_t)8 << 30) / sizeof(T)));
    const auto t0 = std::chrono::steady_clock::now();
    if (wfm_dmalloc((void**)&p, want * sizeof(T)) != hipSuccess) {  // (the cache has given everything back and tried again by then)
      (void)hipGetLastError();  // the failure must not surface after a later launch
      if (wfm_dmalloc((void**)&p, n * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return -1; }
      want = n;
    }
    cap = want;
    if (want * sizeof(T) >= ((size_t)256 << 20) && getenv("WFM_DEBUG"))
      fprintf(stderr, "[wfm] device block of %.2f GB took %.1f ms\n", (double)(want * sizeof(T)) / 1073741824.0,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
  }
  void release() { if (p) wfm_dfree(p); p = nullptr; cap = 0; }
};

struct ProbMeta {
  int64_t p_fwd, t_fwd, p_rev, t_rev;  // offsets into device sequence buffer
  int32_t plen, tlen;
  int32_t mode, pbf, pef, tbf, tef;
  int32_t hint;     // the caller's guess of an upper bound of the score (0: none)
  int64_t rle_off;  // start of this problem's RLE slot range
};

struct Node {
  int32_t prob;
  int32_t pb, pl, tb, tl;
  int32_t cb, ce;
  int32_t score_rem;  // INT_MAX at the root
  int32_t smax;       // base jobs: score budget (0 = derive)
  int32_t endsfree;
  int32_t noband;     // bialign jobs: 1 = ran out of a narrow ring once, gets the full one now
  int32_t sub;        // bialign jobs: upper bound of the score (SUB_NONE: none); the wavefronts are cut to what can stay under it
  int32_t hinted;     // the bound is the caller's guess (a root): the job is run again without it if the guess was too small
  int32_t tries;      // base jobs: how many score budgets the job has overflowed so far
};

}  // namespace

struct wfm_seqset {
  uint8_t* d_seq = nullptr;
  uint32_t* d_pk = nullptr;          // 2-bit mirror of d_seq (wfa_tile2.hip), PK_PAD_WORDS behind it
  std::vector<int32_t> acgt;         // per problem: nonzero = both sequences are pure upper-case ACGT
  size_t bytes = 0;
  std::vector<ProbMeta> meta;
  int64_t rle_total = 0;
  uint64_t seq_bases = 0;
};

namespace {
// one time origin per device for the whole process: the busy intervals of calls on different handles of a device (the align
// driver keeps several batches in flight, each on a handle of its own) are reported against it and can be merged
std::mutex g_base_mu;
hipEvent_t g_dev_base[64] = {};
int g_dev_handles[64] = {};  // live handles per device (wfm_create / wfm_destroy)
// The origin as (event, milliseconds from the process's first origin on that device to the event).  hipEventElapsedTime
// returns a float: against an origin hours old its resolution is a millisecond or worse, so the origin is moved up every few
// minutes and the distance it has moved is kept as a double.
double g_dev_base_off[64] = {};
hipEvent_t device_base_event(int device, double* off_ms) {
  std::lock_guard<std::mutex> lk(g_base_mu);
  if (off_ms) *off_ms = 0;
  if (device < 0 || device >= 64) return nullptr;
  static hipStream_t clock_stream[64] = {};  // (a stream of its own, non-blocking: an event on the null stream would wait for every handle's work)
  if (!clock_stream[device] && hipStreamCreateWithFlags(&clock_stream[device], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); clock_stream[device] = nullptr; }
  hipStream_t cs = clock_stream[device];
  auto fresh = [cs] {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, cs) == hipSuccess && hipEventSynchronize(e) == hipSuccess) return e;
    if (e) (void)hipEventDestroy(e);
    (void)hipGetLastError();
    return (hipEvent_t) nullptr;
  };
  if (!g_dev_base[device]) g_dev_base[device] = fresh();
  else {
    // (the age is taken with an event of the moment: cheap, and only on the rare calls that ask for the origin)
    hipEvent_t now = fresh();
    float age = 0;
    if (now && hipEventElapsedTime(&age, g_dev_base[device], now) == hipSuccess && age > 240000.0f) {
      // (handles that recorded their call's origin against the old event keep working: the old event is left alive)
      g_dev_base_off[device] += (double)age;
      g_dev_base[device] = now;
    } else if (now) (void)hipEventDestroy(now);
  }
  if (off_ms) *off_ms = g_dev_base_off[device];
  return g_dev_base[device];
}
}  // namespace

namespace { void (*g_last_handle_hook)() = nullptr; }
void wfm_set_last_handle_hook(void (*f)()) { g_last_handle_hook = f; }

struct wfm_handle {
  int device = 0;
  std::vector<std::pair<double, double>> busy_abs;  // merged intervals during which a kernel of the last align call ran, ms after the device's origin
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  std::vector<hipEvent_t> tile_ev;  // start/stop pairs for the tile blocks of one chunk
  std::vector<wfm_handle*> peers;   // further contexts for the other parts of a batch (created on first use)
  bool is_peer = false;             // a part's context: it shares its owner's budget and is not counted as a handle of the device
  hipEvent_t ev_base = nullptr;     // time origin of the call (shared by the two halves)
  hipEvent_t call_base = nullptr;   // the origin this call measures against
  std::vector<std::pair<float, float>> tile_iv;  // (start, end) of every tile kernel launch of the call, ms after call_base
  std::vector<std::pair<float, float>> bp_iv, base_iv;  // the same for the step kernel and the base kernel
  std::string err;
  std::string name;
  std::vector<uint32_t> prob_flags;  // WFM_PF_* of every problem of the last align call (wfm_get_problem_flags)
  int other_calls = 0;         // wfm_set_concurrent_calls: align calls the caller keeps in flight on this device beside this handle's
  size_t mem_budget = 0;       // arena budget in force for the call at hand
  size_t mem_budget_full = 0;  // the handle's whole budget (40 % of free HBM at creation, or WFM_MEM_BUDGET_MB)
  wfm_stats_t stats{};
  DevBuf<int32_t> ring;      // breakpoint rings
  DevBuf<int32_t> base32;    // base: pre + rings
  DevBuf<uint8_t> base8;     // base: bt
  DevBuf<uint32_t> rle, rle_out;
  DevBuf<BpJob> bpjobs;
  DevBuf<TileJob> tilejobs;
  DevBuf<TileTask> tiletasks;
  DevBuf<int32_t> tilemak;
  DevBuf<int32_t> p2rows, p2max, p2bmax, p2pbmax;  // phase 2 from rows computed ahead (P2Job)
  DevBuf<P2Job> p2jobs;
  DevBuf<SeqRev> revjobs;
  DevBuf<BoundJob> bndjobs;   // roots whose score is bounded from above before their wavefronts run (wfa_bound_kernel)
  DevBuf<int32_t> bndres;
  uint8_t* stage = nullptr;  // pinned staging buffer of wfm_upload_sequences (grow-only)
  size_t stage_cap = 0;
  DevBuf<BpResult> bpres;
  DevBuf<BaseJob> bsjobs;
  DevBuf<BaseResult> bsres;
  DevBuf<Base2TJob> b2tjobs;     // base jobs on tiles (wfa_base2t_kernel)
  DevBuf<Base2TTask> b2ttasks;
  DevBuf<unsigned long long> b2tkeys;
  DevBuf<int32_t> b2toffs, b2tactive;
  DevBuf<int64_t> i64a, i64b, i64c;
  DevBuf<int32_t> i32a;
  DevBuf<int32_t> seqflags;  // wfm_upload_sequences: per problem, nonzero = pure ACGT
  DevBuf<SeqRev> flagjobs;
  DevBuf<unsigned long long> total;
  void* attachment = nullptr;  // owned by another translation unit (map_kernels.hip: the pinned staging ring)
  void (*attachment_free)(void*) = nullptr;
};

namespace {

inline int gapcost(const wfm_penalties_t& p, int L) {
  if (L <= 0) return 0;
  return std::min(p.o1 + p.e1 * L, p.o2 + p.e2 * L);
}

// Row ranges with a bound of the score (wfa_kernels.hip: Rng): the kernels' arithmetic, for tile counts and cell counts
inline int h_rng_lo(int pl, int tl, int sub, int s) { return std::max(std::max(-pl, -s), (tl - pl) - sub + s); }
inline int h_rng_hi(int pl, int tl, int sub, int s) { return std::min(std::min(tl, s), (tl - pl) + sub - s); }
inline int64_t h_row_cells(int pl, int tl, int sub, int s) { return std::max(0, h_rng_hi(pl, tl, sub, s) - h_rng_lo(pl, tl, sub, s) + 1); }
// sum of h_row_cells over the scores a .. b: the row's edges are piecewise linear in the score (each a min / max of three
// lines), so between two consecutive kinks the count is an arithmetic series
inline int64_t h_cells_sum(int pl, int tl, int sub, int a, int b) {
  if (b < a) return 0;
  const int64_t kinv = (int64_t)tl - pl, khi = kinv + sub, klo = kinv - sub;
  // scores at which two of the lines of an edge cross (the kink lies between the floor and the next integer)
  int64_t cand[16];
  int nc = 0;
  auto add = [&](int64_t x) { for (int64_t y : {x, x + 1}) if (y > a && y <= b) cand[nc++] = y; };
  add(tl); add(khi / 2 - (khi < 0 && (khi & 1) ? 1 : 0)); add(khi - tl);     // hi: s vs tl, s vs khi - s, tl vs khi - s
  add(pl); add((-klo) / 2 - (-klo < 0 && ((-klo) & 1) ? 1 : 0)); add(-klo - pl);  // lo: -s vs -pl, -s vs klo + s, -pl vs klo + s
  std::sort(cand, cand + nc);
  int64_t total = 0;
  int64_t u = a;
  auto cells = [&](int64_t s) { return (int64_t)h_rng_hi(pl, tl, sub, (int)s) - h_rng_lo(pl, tl, sub, (int)s) + 1; };
  auto seg = [&](int64_t x, int64_t y) {  // linear on [x, y]
    if (y < x) return;
    const int64_t cx = cells(x), cy = cells(y);
    if (cx <= 0 && cy <= 0) return;
    if (cx > 0 && cy > 0) { total += (cx + cy) * (y - x + 1) / 2; return; }
    if (y == x) { total += std::max<int64_t>(cx, 0); return; }
    // one end at or below zero: the slope is (cy - cx) / (y - x), an integer (each edge moves by whole diagonals per score)
    const int64_t slope = (cy - cx) / (y - x);
    if (cx > 0) {  // falls: positive up to x + (cx - 1) / -slope
      const int64_t last = x + (cx - 1) / (-slope);
      total += (cx + cells(last)) * (last - x + 1) / 2;
    } else {       // rises: positive from y - (cy - 1) / slope
      const int64_t first = y - (cy - 1) / slope;
      total += (cells(first) + cy) * (y - first + 1) / 2;
    }
  };
  for (int q = 0; q < nc; ++q) {
    if (cand[q] <= u) continue;
    seg(u, cand[q] - 1);
    u = cand[q];
  }
  seg(u, b);
  return total;
}
// (the kernels' rng_block: the score bound as it stood RNG_BACK = 25 scores before the block, see wfa_kernels.hip)
inline void h_rng_block(int pl, int tl, int sub, int s_from, int s_to, int* L, int* R) {
  *L = std::max(std::max(-pl, -s_to), (tl - pl) - sub + s_from - 25);
  *R = std::min(std::min(tl, s_to), (tl - pl) + sub - s_from + 25);
}

int validate_pen(const wfm_penalties_t* pen, int* scope) {
  if (!pen) return WFM_E_ARG;
  if (pen->x <= 0 || pen->e1 <= 0 || pen->e2 <= 0 || pen->o1 < 0 || pen->o2 < 0) return WFM_E_UNSUPPORTED;
  const int sc = std::max(pen->x, std::max(pen->o1 + pen->e1, pen->o2 + pen->e2)) + 1;
  // (two rows of a ring are always in the making: the step kernel clears the row-maximum slot of row s + 2 while rows back to
  // s - scope + 1 are still read, so a ring of R rows serves scopes up to R - 2)
  if (sc > RING_BIG - 2) return WFM_E_UNSUPPORTED;  // o2 + e2 (or o1 + e1, or x) beyond 125: deeper rings than the kernels are built with
  *scope = sc;
  return WFM_OK;
}

// rows of a wavefront ring for penalties of this scope: 32 (the default penalties: scope 26) or 128
inline int ring_rows_for(int scope) { return scope <= RING - 2 ? RING : RING_BIG; }

struct LevelTimer {
  double bp_ms = 0, base_ms = 0, tile_ms = 0;
};

// widest row a base job can reach (0 for the trivial all-gap jobs)
// The diagonals a base job's rows have to hold under its score budget: within `smax` of where an alignment may begin (diagonal 0 of an
// end-to-end job, [-pbf, tbf] of an ends-free one) -- and, when the alignment has to END in the far corner (no free ends there: leaves, and the
// head patches, whose free ends are at the beginning), within `smax` of the corner's diagonal tl - pl as well: every change of diagonal costs at least
// e2 = 1, so a cell further away lies on no alignment of score <= smax, no cell on such an alignment takes its value from one (the argument of the
// score bounds, section 5 of DESIGN.md), and a job that needs more than its budget is run again anyway.  A head patch begins with ALL its diagonals
// (its begin-free lengths are the eroded lengths: rows of 2 - 8 k diagonals for a budget of 256); with the corner's band its rows are 513 wide.
inline void base_columns(const Node& nd, const ProbMeta& pm, int64_t* kmin_out, int64_t* kmax_out) {
  int64_t kmin = nd.endsfree ? std::max<int64_t>(-nd.pl, -(int64_t)pm.pbf - nd.smax) : std::max<int64_t>(-nd.pl, -nd.smax);
  int64_t kmax = nd.endsfree ? std::min<int64_t>(nd.tl, (int64_t)pm.tbf + nd.smax) : std::min<int64_t>(nd.tl, nd.smax);
  static const bool corner_band = !(getenv("WFM_BASE_CORNER_BAND") && atoi(getenv("WFM_BASE_CORNER_BAND")) == 0);
  const bool end_fixed = !nd.endsfree || (pm.pef == 0 && pm.tef == 0);
  if (corner_band && end_fixed) {
    const int64_t k_end = (int64_t)nd.tl - nd.pl;
    const int64_t bmin = std::max(kmin, k_end - nd.smax), bmax = std::min(kmax, k_end + nd.smax);
    // the first row must keep a cell inside (a budget that cannot reach the corner at all leaves the columns as they were: the job overflows as before)
    const int64_t lo0 = nd.endsfree ? std::max<int64_t>(-(int64_t)pm.pbf, bmin) : 0, hi0 = nd.endsfree ? std::min<int64_t>((int64_t)pm.tbf, bmax) : 0;
    if (bmin <= bmax && lo0 <= hi0 && lo0 >= bmin && hi0 <= bmax) { kmin = bmin; kmax = bmax; }
  }
  *kmin_out = kmin; *kmax_out = kmax;
}
inline int64_t base_row_width(const Node& nd, const ProbMeta& pm) {
  if (nd.tl == 0 || nd.pl == 0) return 0;
  int64_t kmin, kmax;
  base_columns(nd, pm, &kmin, &kmax);
  return kmax - kmin + 1;
}

// Runs all base jobs of `nodes` (chunked to the memory budget, wide jobs apart from narrow ones); appends
// overflowed nodes (with a larger budget) to `retry`.
int run_base_jobs(wfm_handle* h, wfm_seqset* S, const wfm_penalties_t& pen, std::vector<Node>& nodes,
                  std::vector<Node>& retry, std::vector<int32_t>& prob_status, std::vector<uint64_t>& prob_cells,
                  LevelTimer& tm, uint32_t* pflags) {
  if (nodes.empty()) return WFM_OK;
  const int RR = ring_rows_for(std::max(pen.x, std::max(pen.o1 + pen.e1, pen.o2 + pen.e2)) + 1);
  // rows beyond 2 k diagonals get 1024 threads -- and rows beyond 512 when the launch is too small to fill the device anyway
  // (the retries of the few patches that overflowed their first budget: one workgroup each, a thousand steps deep)
  const int wide_from = nodes.size() < 128 ? 512 : 2048;
  // Kinds of jobs, each in launches of its own: 0 / 1 / 2 = the register kernel on packed sequences (wfa_base2_kernel: default
  // penalties, pure ACGT, rows up to 128 / 512 / 2048 diagonals, sequences that fit its windows), 3 / 4 = the ring kernel with
  // 256 / 1024 threads (other penalties, an N, wider rows: a patch eroded to its 4096-base limit starts 8 k diagonals wide)
  const bool dflt_pen = pen.x == 5 && pen.o1 == 8 && pen.e1 == 2 && pen.o2 == 24 && pen.e2 == 1;
  const bool base_v2 = dflt_pen && !(getenv("WFM_BASE_V2") && atoi(getenv("WFM_BASE_V2")) == 0) && !(getenv("WFM_TILE_V2") && atoi(getenv("WFM_TILE_V2")) == 0);
  // 5 = the register kernel's step on tiles (wfa_base2t_kernel): rows beyond 2048 diagonals of jobs the register kernel would take -- the third
  // attempt of a patch, whose score passed 1020
  const bool base_tiles = !(getenv("WFM_BASE_TILES") && atoi(getenv("WFM_BASE_TILES")) == 0);
  const bool force_tiles = getenv("WFM_BASE_TILES") && atoi(getenv("WFM_BASE_TILES")) == 2;  // tests: every leaf and patch with rows beyond 128 diagonals
  // (sequences longer than the kernel's LDS windows are fine: what lies beyond is read from the global mirror.  Jobs without
  // any cell -- an empty pattern or text -- ride along with the first kind: they are one store each)
  const bool few_jobs = nodes.size() < 128;
  auto kind_of = [&](const Node& a) {
    const int64_t w = base_row_width(a, S->meta[a.prob]);
    if (base_v2 && (a.pl == 0 || a.tl == 0)) return 1;
    if (base_v2 && w <= 2048 && (size_t)a.prob < S->acgt.size() && S->acgt[(size_t)a.prob])
      // (a handful of retries: more workgroups of fewer waves per job on the tiles of the register kernel, and ONE launch with the wider ones
      // instead of one per width class, each a few jobs and hundreds of score steps long)
      return w <= 128 ? 0 : (base_tiles && (force_tiles || (few_jobs && (a.tries > 0 || w > 640))) ? 5 : (w <= 640 ? 1 : 2));
    if (base_v2 && base_tiles && (size_t)a.prob < S->acgt.size() && S->acgt[(size_t)a.prob]) return 5;
    return w > wide_from ? 4 : 3;
  };
  std::stable_sort(nodes.begin(), nodes.end(), [&](const Node& a, const Node& b) { return kind_of(a) < kind_of(b); });
  const DevPen dp{pen.x, pen.o1, pen.e1, pen.o2, pen.e2};
  size_t i0 = 0;
  std::vector<BaseJob> jobs;
  std::vector<BaseResult> res;
  while (i0 < nodes.size()) {
    jobs.clear();
    size_t n32 = 0, n8 = 0;
    size_t i = i0;
    int chunk_kind = 3;
    for (; i < nodes.size(); ++i) {
      const Node& nd = nodes[i];
      const ProbMeta& pm = S->meta[nd.prob];
      {  // a chunk holds jobs of one kind
        const int kind = kind_of(nd);
        if (jobs.empty()) chunk_kind = kind;
        else if (kind != chunk_kind) break;
      }
      BaseJob j{};
      j.p_off = pm.p_fwd + nd.pb;
      j.t_off = pm.t_fwd + nd.tb;
      j.pl = nd.pl; j.tl = nd.tl;
      j.comp_begin = nd.cb; j.comp_end = nd.ce;
      j.endsfree = nd.endsfree;
      j.pbf = pm.pbf; j.pef = pm.pef; j.tbf = pm.tbf; j.tef = pm.tef;
      j.rle_end = pm.rle_off + nd.pb + nd.tb + nd.pl + nd.tl;
      j.pad_ = (int32_t)i;
      if (nd.tl == 0 || nd.pl == 0) {
        j.type = nd.tl == 0 ? 1 : 2;
        if (nd.tl == 0 && nd.pl == 0) { j.type = 1; }
        jobs.push_back(j);
        continue;
      }
      j.type = 0;
      j.smax = nd.smax;
      int64_t kmin64, kmax64;
      base_columns(nd, pm, &kmin64, &kmax64);
      const int kmin = (int)kmin64, kmax = (int)kmax64;
      j.kmin = kmin;
      j.width = kmax - kmin + 1;
      const size_t rows = (size_t)nd.smax + 1;
      const size_t need32 = rows * (size_t)j.width + (size_t)5 * RR * (size_t)j.width;
      const size_t need8 = rows * (size_t)j.width;
      // (a chunk of base jobs stops at 4 GB of arenas even where the budget allows more, like a chunk of rings: with the leaves of all levels going out
      // together a batch of divergent records asked for 18 GB blocks -- 0.6 s each as a first allocation, gpurun_out/r5u_c1.err -- and thousands of
      // leaves fill the device long before that)
      static const size_t base_chunk_bytes = (size_t)(getenv("WFM_BASE_CHUNK_GB") ? std::max(1, atoi(getenv("WFM_BASE_CHUNK_GB"))) : 4) << 30;
      if (!jobs.empty() && (n32 + need32) * 4 + (n8 + need8) > std::min(h->mem_budget, base_chunk_bytes)) break;
      if (need32 * 4 + need8 > h->mem_budget) {  // a single job beyond the budget
        prob_status[nd.prob] = WFM_ST_OOM;
        continue;
      }
      j.pre_off = (int64_t)n32;
      j.ring_off = (int64_t)(n32 + rows * (size_t)j.width);
      j.bt_off = (int64_t)n8;
      n32 += need32; n8 += need8;
      jobs.push_back(j);
    }
    const size_t chunk_end = i;
    if (!jobs.empty()) {
      if (h->base32.ensure(n32 + 16) || h->base8.ensure(n8 + 16) || h->bsjobs.ensure(jobs.size()) || h->bsres.ensure(jobs.size())) {
        h->err = "out of device memory (base arena)";
        return WFM_E_NOMEM;
      }
      HIPCHK(h, hipMemcpyAsync(h->bsjobs.p, jobs.data(), jobs.size() * sizeof(BaseJob), hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipEventRecord(h->ev2, h->stream));
      // (jobs arrive sorted: the wide ones -- long patches, retries with a larger budget -- in chunks of their own)
      const bool chunk_wide = chunk_kind == 4;
      if (chunk_kind == 5) {
        // blocks of T scores, every block one launch over the tiles of all jobs and a one-thread-per-job kernel behind it; the host looks at the
        // number of jobs still running every few blocks (a launch whose jobs are all over costs microseconds)
        static const int T = getenv("WFM_BASE_TILE_T") ? std::max(5, std::min(400, atoi(getenv("WFM_BASE_TILE_T")) / 5 * 5)) : 125;
        const int core = B2T_THREADS * 2 - 2 * T;
        std::vector<Base2TJob> tj(jobs.size());
        std::vector<Base2TTask> tasks;
        int smax_all = 0;
        for (size_t q = 0; q < jobs.size(); ++q) {
          Base2TJob& t = tj[q];
          t.b = jobs[q];
          t.snap_in = jobs[q].ring_off; t.snap_out = jobs[q].ring_off + (int64_t)B2T_ROWS * jobs[q].width;
          t.core = core; t.ntiles = (jobs[q].width + core - 1) / core; t.task0 = (int32_t)tasks.size();
          t.s0 = 0; t.done = 0; t.end_s = 0; t.end_k = 0; t.end_off = 0;
          for (int ti = 0; ti < t.ntiles; ++ti) tasks.push_back(Base2TTask{(int32_t)q, ti});
          smax_all = std::max(smax_all, jobs[q].smax);
        }
        const int nblocks = (smax_all + T - 1) / T + 1;
        if (h->b2tjobs.ensure(tj.size()) || h->b2ttasks.ensure(tasks.size()) || h->b2tkeys.ensure(tasks.size()) || h->b2toffs.ensure(tasks.size()) || h->b2tactive.ensure((size_t)nblocks)) {
          h->err = "out of device memory (base tiles)";
          return WFM_E_NOMEM;
        }
        HIPCHK(h, hipMemcpyAsync(h->b2tjobs.p, tj.data(), tj.size() * sizeof(Base2TJob), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->b2ttasks.p, tasks.data(), tasks.size() * sizeof(Base2TTask), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->b2tactive.p, 0, (size_t)nblocks * sizeof(int32_t), h->stream));
        constexpr int LOOK = 6;
        for (int b = 0; b < nblocks; ) {
          const int upto = std::min(nblocks, b + LOOK);
          for (; b < upto; ++b) {
            launch_base2t_block(S->d_pk, h->base32.p, h->base8.p, h->b2tjobs.p, h->b2ttasks.p, h->b2tkeys.p, h->b2toffs.p, (int)tasks.size(), T, h->stream);
            launch_base2t_advance(h->b2tjobs.p, h->b2tkeys.p, h->b2toffs.p, (int)tj.size(), T, h->b2tactive.p + b, h->stream);
          }
          int32_t still = 0;
          HIPCHK(h, hipMemcpyAsync(&still, h->b2tactive.p + (b - 1), sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
          HIPCHK(h, hipStreamSynchronize(h->stream));
          if (!still) break;
        }
        launch_base2t_finish(h->base32.p, h->base8.p, h->rle.p, h->b2tjobs.p, h->bsres.p, (int)tj.size(), h->stream);
      } else if (chunk_kind <= 2) {
        int64_t wmax = 1;
        for (const BaseJob& bj : jobs) if (bj.type == 0) wmax = std::max<int64_t>(wmax, bj.width);
        const int threads = (int)std::min<int64_t>(1024, ((wmax + 1) / 2 + 63) / 64 * 64);
        launch_base2(S->d_pk, h->base32.p, h->base8.p, h->rle.p, h->bsjobs.p, h->bsres.p, (int)jobs.size(), threads, h->stream);
      } else
        launch_base(S->d_seq, h->base32.p, h->base8.p, h->rle.p, h->bsjobs.p, h->bsres.p, (int)jobs.size(), dp, chunk_wide, RR, h->stream);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipEventRecord(h->ev3, h->stream));
      res.resize(jobs.size());
      HIPCHK(h, hipMemcpyAsync(res.data(), h->bsres.p, jobs.size() * sizeof(BaseResult), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      float ms = 0;
      HIPCHK(h, hipEventElapsedTime(&ms, h->ev2, h->ev3));
      tm.base_ms += ms;
      if (h->call_base) {
        float t0 = 0;
        HIPCHK(h, hipEventElapsedTime(&t0, h->call_base, h->ev2));
        h->base_iv.emplace_back(t0, t0 + ms);
      }
      h->stats.base_launches++;
      h->stats.base_jobs += (uint32_t)jobs.size();
      if (getenv("WFM_DEBUG") && atoi(getenv("WFM_DEBUG")) > 1) {
        int64_t wsum = 0, wmax = 0, smx = 0; int over = 0, ef = 0;
        for (size_t q = 0; q < jobs.size(); ++q) { wsum += jobs[q].width; wmax = std::max<int64_t>(wmax, jobs[q].width); smx = std::max<int64_t>(smx, jobs[q].smax); over += res[q].status == WFM_DEV_OVERFLOW; ef += jobs[q].endsfree; }
        fprintf(stderr, "[wfm] base launch: %zu jobs (%d ends-free), %d threads, rows %lld wide on average (max %lld), score budget up to %lld, %.3f ms, %d overflowed\n", jobs.size(), ef,
                chunk_kind == 5 ? B2T_THREADS : (chunk_wide ? 1024 : 256), (long long)(wsum / (int64_t)jobs.size()), (long long)wmax, (long long)smx, ms, over);
        if (chunk_kind <= 2) {
          double fw = 0, bk = 0; int fwm = 0, bkm = 0, scm = 0; double scs = 0;
          for (size_t q = 0; q < jobs.size(); ++q) { const int f = (res[q].pad_ >> 16) & 0xffff, b = res[q].pad_ & 0xffff; fw += f; bk += b; fwm = std::max(fwm, f); bkm = std::max(bkm, b); scs += res[q].score; scm = std::max(scm, res[q].score); }
          fprintf(stderr, "[wfm]   register kernel (kind %d): forward %.0f us on average (max %d), walk back %.0f us (max %d), score %.0f on average (max %d)\n", chunk_kind, fw / jobs.size(), fwm, bk / jobs.size(), bkm, scs / jobs.size(), scm);
        }
        else {
          double scs = 0; int scm = 0, scn = INT_MAX;
          for (size_t q = 0; q < jobs.size(); ++q) { scs += res[q].score; scm = std::max(scm, res[q].score); scn = std::min(scn, res[q].score); }
          fprintf(stderr, "[wfm]   %s (kind %d): score %.0f on average (min %d, max %d)\n", chunk_kind == 5 ? "register kernel on tiles" : "ring kernel", chunk_kind, scs / jobs.size(), scn, scm);
        }
      }
      for (size_t q = 0; q < jobs.size(); ++q) {
        const Node& nd = nodes[(size_t)jobs[q].pad_];
        const BaseResult& r = res[q];
        prob_cells[nd.prob] += r.cells;
        h->stats.cells_base += r.cells;
        if (pflags && (chunk_kind == 3 || chunk_kind == 4) && jobs[q].type == 0) pflags[nd.prob] |= WFM_PF_RING_KERNEL;
        if (pflags && chunk_kind == 5) pflags[nd.prob] |= WFM_PF_BASE_TILES;
        if (r.status == WFM_DEV_OVERFLOW) {
          if (pflags) pflags[nd.prob] |= nd.tries == 0 ? WFM_PF_BASE_RETRY : WFM_PF_BASE_RETRY2;
          Node again = nd;
          again.tries = nd.tries + 1;
          // No alignment costs more than the all-gap one.  A job whose begin or end component is a gap state is held to a
          // PIECE there (a BiWFA child that ends inside a D2 gap pays o2 + e2 per base for it, however short it is --
          // its parent counted that gap's opening on the other side of the breakpoint, so the child's own forward score
          // exceeds the score_rem it was handed): the bound takes the dearer piece for both gaps then.
          const bool constrained = nd.cb != C_M || nd.ce != C_M;
          const int64_t bound = constrained
              ? (int64_t)2 * std::max(pen.o1, pen.o2) + (int64_t)std::max(pen.e1, pen.e2) * ((int64_t)nd.pl + nd.tl) + 8
              : (int64_t)gapcost(pen, nd.pl) + gapcost(pen, nd.tl) + 8;
          if (nd.smax >= bound) {
            if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] problem %d: base job pl %d tl %d cb %d ce %d overflowed its score bound %d\n", nd.prob, nd.pl, nd.tl, nd.cb, nd.ce, nd.smax);
            prob_status[nd.prob] = WFM_ST_UNREACHABLE; continue;
          }
          // (x 8; it was x 2 until round 3 and x 4 for a while: every retry is a launch that a few jobs hold up, a budget that is too large costs
          // memory only -- 20 MB for a 2 k-wide patch at 2 k scores -- and a patch that passed 256 is as likely to need 1500 as 500)
          again.smax = (int32_t)std::min<int64_t>((int64_t)nd.smax * 8 + 32, bound);
          // ... but the second attempt stops at 1020: with the band around the end corner's diagonal the rows of a job with that budget are at most
          // 2041 diagonals wide and fit the register kernel (wfa_base2_kernel: 2048), where a budget of 2080 put two dozen patches of an LPA batch
          // on the ring kernel with rows of 2.6 - 3.4 k diagonals (9.7 ms of the batch's 83); the few that need more take a third attempt
          // (round 6: a job the tiles of the register kernel take -- wfa_base2t_kernel, rows of any width -- has no use for the stop: its second attempt
          // runs with the eightfold budget at once, 2.0 ms of an LPA batch's patch chain less)
          const bool to_tiles = base_v2 && base_tiles && (size_t)nd.prob < S->acgt.size() && S->acgt[(size_t)nd.prob];
          if (nd.smax < 1020 && again.smax > 1020 && !to_tiles) again.smax = 1020;
          retry.push_back(again);
        } else if (r.status != 0) {
          if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] problem %d: base job pl %d tl %d status %d\n", nd.prob, nd.pl, nd.tl, r.status);
          prob_status[nd.prob] = WFM_ST_UNREACHABLE;
        }
      }
    }
    i0 = chunk_end;
  }
  return WFM_OK;
}


struct TileCfg {
  bool exact = true;  // register tiles: re-run the block that holds the meeting point up to that point only, so the step
                      // kernel starts at phase 2 (WFM_TILE_EXACT=0: it redoes the block step by step instead)
  int T_refine = 0;   // optional second pass over the block that holds the meeting point, in finer blocks (WFM_TILE_T_REFINE;
                      // measured neutral on C3: what the step kernel saves, the small tiles cost)
  int chunk = 2;  // tile blocks launched back to back between two looks of the host (WFM_TILE_CHUNK); more only adds idle tiles
  int T = 100, Wt = 1024, threads = 512;  // T: scores per tile block (measured optimum 96-100 on C3: halo 2T of 1024 columns vs per-tile snapshot cost)
  int min_len = 128, min_score = 64;  // shorter problems go to the step kernel whole (they are base jobs at <= 100 anyway).  600 until round 2: the
                                      // short, high-score children a 1 kb end gap leaves behind then queued up as one-workgroup step jobs
  bool enabled = true;
  bool reg = false;  // register-resident tile kernel (default penalty lags only)
  int C = 2;
};

TileCfg tile_cfg(const wfm_penalties_t& pen, int scope) {
  TileCfg c;
  if (const char* e = getenv("WFM_TILE")) c.enabled = atoi(e) != 0;
  if (const char* e = getenv("WFM_TILE_T")) c.T = atoi(e);
  if (const char* e = getenv("WFM_TILE_T_REFINE")) c.T_refine = atoi(e);
  if (const char* e = getenv("WFM_TILE_EXACT")) c.exact = atoi(e) != 0;
  if (const char* e = getenv("WFM_TILE_W")) c.Wt = atoi(e);
  if (const char* e = getenv("WFM_TILE_THREADS")) c.threads = atoi(e);
  if (const char* e = getenv("WFM_TILE_MIN_LEN")) c.min_len = atoi(e);
  if (const char* e = getenv("WFM_TILE_MIN_SCORE")) c.min_score = atoi(e);
  const int RR = ring_rows_for(scope);
  c.T = std::max(c.T, RR);  // the output snapshot needs `scope` rows of the block itself
  if (c.T_refine > 0) c.T_refine = std::max(c.T_refine, RR);
  const bool dflt = pen.x == 5 && pen.o1 + pen.e1 == 10 && pen.o2 + pen.e2 == 25 && pen.e1 == 2 && pen.e2 == 1;
  c.reg = dflt && !(getenv("WFM_TILE_REG") && atoi(getenv("WFM_TILE_REG")) == 0);
  if (c.reg) {
    if (!getenv("WFM_TILE_THREADS")) c.threads = 512;
    // (the FINE instantiation of the packed kernel keeps a row of T + 1 maxima per wave in dynamic LDS beside ~24 KB of static: a WFM_TILE_T that
    // would not fit 64 KB is cut here instead of failing the launch)
    c.T = std::min(c.T, (int)((40 * 1024) / (4 * std::max(1, c.threads / 64))) - 1);
    // (two diagonals per lane, always: the phase-2 rows, the single-tile sizing and the packed kernel are built for it.  The WFM_TILE_C=4
    // switch of round 1 sized the tasks for four while those stages went on with two -- wrong results; it is gone)
    if (const char* e = getenv("WFM_TILE_CHUNK")) c.chunk = std::max(1, atoi(e));
    c.Wt = c.threads * c.C;
    return c;
  }
  const size_t rows = (size_t)scope + 2 * (pen.e1 + 1) + 2 * (pen.e2 + 1);
  while ((rows * c.Wt + c.T + 1) * 4 > 160 * 1024 && c.Wt > 4 * c.T) c.Wt -= 64;
  if (c.Wt < 4 * c.T || (rows * c.Wt + c.T + 1) * 4 > 160 * 1024) c.enabled = false;  // (a scope beyond ~60 rows: the step kernel does it all)
  return c;
}

// Advances the jobs listed in `tiled` (indices into jobs) in blocks of T scores with the
// time-tiled kernel until their forward/reverse antidiagonals meet inside a block; then
// leaves them positioned at the last snapshot for wfa_bp_kernel (resume_s/fmax0/rmax0/ring_off).
// With `refine` the jobs continue from where a coarser pass left them (blocks of T scores again, T smaller),
// so that the step-by-step kernel has at most the finer T steps to redo.
int run_tiled_phase(wfm_handle* h, wfm_seqset* S, const DevPen& dp, int scope, const TileCfg& cfg, int T, bool refine,
                    std::vector<BpJob>& jobs, const std::vector<int>& tiled, std::vector<int64_t>& ring2,
                    double& tile_ms, uint64_t& tile_cells, uint32_t level, const std::vector<int32_t>* fine_from = nullptr,
                    const std::vector<int64_t>* ring3 = nullptr) {
  const size_t n = tiled.size();
  if (n == 0) return WFM_OK;
  double lane_cells = 0;  // threads x diagonals per thread x scores over all tiles launched (diagnostics)
  const double cells_before = (double)tile_cells;
  const int core = cfg.Wt - 2 * T;
  std::vector<TileJob> tj(n);
  std::vector<int> fmax(n, 0), rmax(n, 0);
  std::vector<char> active(n, 1);
  for (size_t i = 0; i < n; ++i) {
    const BpJob& j = jobs[(size_t)tiled[i]];
    TileJob& t = tj[i];
    t.p_fwd = j.p_fwd; t.t_fwd = j.t_fwd; t.p_rev = j.p_rev; t.t_rev = j.t_rev;
    t.ring_in = j.ring_off; t.ring_out = ring2[i];
    t.pl = j.pl; t.tl = j.tl; t.comp_begin = j.comp_begin; t.comp_end = j.comp_end;
    t.width = j.width; t.koff = j.koff; t.s0 = 0; t.active = 1; t.fmax = 0; t.rmax = 0; t.nblocks = 0; t.mode = 0; t.tf = 0; t.tr = 0; t.last_fwd = 0; t.packed = j.packed;
    t.p2_off = 0; t.w2 = 0; t.koff2 = 0; t.sub = j.sub;
    t.fine_s = (fine_from && i < fine_from->size()) ? (*fine_from)[i] : INT_MAX;  // (TileJob::fine_s; a job whose score nobody knows finds its meeting block by running it again)
    // a third ring (TileJob::ring_prev): packed jobs of the exact tile phase only -- the byte kernel and the step kernel's fall-backs read gap rows
    // 26 deep from any snapshot
    t.ring_prev = (ring3 && i < ring3->size() && (*ring3)[i] >= 0 && (j.packed & 1) && cfg.reg && cfg.exact && !refine) ? (*ring3)[i] : -1;
    t.prev_ok = 0; t.reran = 0;
  }
  bool any_cut = false;  // the kernel form with the score bounds' bookkeeping is only launched when a job carries one
  for (size_t i = 0; i < n; ++i) any_cut |= tj[i].sub != SUB_NONE;
  if (h->tilejobs.ensure(n) || h->tilemak.ensure(n * 2 * (size_t)std::max(T, 2))) { h->err = "out of device memory (tiles)"; return WFM_E_NOMEM; }
  size_t n_active = 0;
  std::vector<int> s_begin(n, 0);  // score the jobs start this pass at
  if (!refine) {
    HIPCHK(h, hipMemcpyAsync(h->tilejobs.p, tj.data(), n * sizeof(TileJob), hipMemcpyHostToDevice, h->stream));
    launch_tile_init(S->d_seq, h->ring.p, h->tilejobs.p, h->tilemak.p, (int)n, ring_rows_for(scope), h->stream);
    HIPCHK(h, hipGetLastError());
    std::vector<int32_t> mak(n * 4);
    HIPCHK(h, hipMemcpyAsync(mak.data(), h->tilemak.p, n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < n; ++i) {
      fmax[i] = mak[(i * 2 + 0) * 2]; rmax[i] = mak[(i * 2 + 1) * 2];
      const bool ended = mak[(i * 2 + 0) * 2 + 1] || mak[(i * 2 + 1) * 2 + 1];
      const int A = tj[i].pl + tj[i].tl - 1;
      if (ended || fmax[i] + rmax[i] >= A) active[i] = 0;  // wfa_bp_kernel handles it from score 0
      n_active += active[i];
      tj[i].active = active[i]; tj[i].fmax = fmax[i]; tj[i].rmax = rmax[i]; tj[i].nblocks = 0;
    }
  } else {
    for (size_t i = 0; i < n; ++i) {
      const BpJob& j = jobs[(size_t)tiled[i]];
      const int A = tj[i].pl + tj[i].tl - 1;
      fmax[i] = j.fmax0; rmax[i] = j.rmax0;
      s_begin[i] = j.resume_s;
      active[i] = (char)(fmax[i] + rmax[i] < A);  // jobs that were over before their first block stay where they are
      n_active += active[i];
      tj[i].s0 = j.resume_s; tj[i].active = active[i]; tj[i].fmax = fmax[i]; tj[i].rmax = rmax[i]; tj[i].nblocks = 0;
    }
  }
  const size_t lds = ((size_t)(scope + 2 * (dp.e1 + 1) + 2 * (dp.e2 + 1)) * cfg.Wt + T + 1) * 4;
  uint32_t blocks = 0;
  if (n_active) {
    // The per-job state lives on the device and a tiny kernel between two blocks replays the termination
    // checks, so the host only looks every `chunk` blocks.  The task list is built per chunk for the widest
    // range the chunk can reach; a block's tiles outside its current range, and all tiles of jobs that
    // finished earlier in the chunk, exit at once.
    std::vector<TileTask> tasks, tasks_by;
    HIPCHK(h, hipMemcpyAsync(h->tilejobs.p, tj.data(), n * sizeof(TileJob), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->tilemak.p, 0, n * 2 * (size_t)T * sizeof(int32_t), h->stream));
    const int chunk = std::max(1, std::min(cfg.chunk, (int)h->tile_ev.size() / 2));
    std::vector<TileJob> got(n);
    auto clk = [] { return std::chrono::steady_clock::now(); };
    auto msd = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    double ms_prep = 0, ms_launch = 0, ms_wait = 0, ms_post = 0;
    while (n_active) {
      const auto tq0 = clk();
      // a job on a narrow ring (BpJob::band) may only start a chunk whose last score still fits; otherwise it leaves
      // the tile phase here and is run again on a full ring (the host retries it)
      bool out_of_band = false;
      for (size_t i = 0; i < n; ++i) {
        const int band = jobs[(size_t)tiled[i]].band;
        if (active[i] && band > 0 && tj[i].s0 + chunk * T + 2 > band) {
          active[i] = 0; tj[i].active = 0; tj[i].mode = 3; out_of_band = true; --n_active;
        }
        // a job whose score bound is a guess: the two directions meet near half the score, so one that is still going
        // well past half the bound has a score above it -- it leaves here as well and is run again without the bound
        if (active[i] && tj[i].sub != SUB_NONE) {
          int L, R;
          h_rng_block(tj[i].pl, tj[i].tl, tj[i].sub, tj[i].s0, tj[i].s0 + T, &L, &R);
          if (2 * tj[i].s0 > tj[i].sub + 128 || R < L) {  // (or nothing is left within the bound)
            active[i] = 0; tj[i].active = 0; tj[i].mode = 3; out_of_band = true; --n_active;
          }
        }
      }
      if (out_of_band) HIPCHK(h, hipMemcpyAsync(h->tilejobs.p, tj.data(), n * sizeof(TileJob), hipMemcpyHostToDevice, h->stream));
      if (!n_active) break;
      // as many tiles of `core` diagonals per job and direction as the last block of this chunk can need.  While the
      // widest range of the chunk fits one tile, every job-direction is ONE tile without a halo, and the workgroups
      // are only as large as that range needs (the first blocks of a level are a few hundred diagonals wide)
      tasks.clear();
      tasks_by.clear();
      int core_c = core;
      std::vector<int> threads_b((size_t)chunk, cfg.threads);  // workgroup size of every block of the chunk
      if (cfg.reg && cfg.C == 2) {
        std::vector<int> wb((size_t)chunk, 0);  // widest range per block
        for (size_t i = 0; i < n; ++i) {
          if (!active[i]) continue;
          // a job may run the same block twice (the block in which its wavefronts met, up to the meeting point), and with a
          // score bound the ranges shrink again towards the end: block b of the chunk needs the widest range up to b
          int wmax = 0;
          for (int b = 0; b < chunk; ++b) {
            int L, R;
            h_rng_block(tj[i].pl, tj[i].tl, tj[i].sub, tj[i].s0 + b * T, tj[i].s0 + (b + 1) * T, &L, &R);
            wmax = std::max(wmax, R - L + 1);
            wb[(size_t)b] = std::max(wb[(size_t)b], wmax);
          }
        }
        if (wb[(size_t)chunk - 1] <= cfg.threads * cfg.C) {
          // one tile per job-direction: whole waves, as many as the block's own widest range needs (two diagonals per lane)
          static const bool fine = !(getenv("WFM_TILE_FINE") && atoi(getenv("WFM_TILE_FINE")) == 0);
          for (int b = 0; b < chunk; ++b) {
            const int wdt = fine ? wb[(size_t)b] : wb[(size_t)chunk - 1];
            threads_b[(size_t)b] = fine ? std::min(cfg.threads, std::max(64, ((wdt + cfg.C - 1) / cfg.C + 63) / 64 * 64))
                                        : std::min(cfg.threads, wdt <= 256 ? 128 : (wdt <= 512 ? 256 : cfg.threads));
          }
          core_c = threads_b[(size_t)chunk - 1] * cfg.C;
        }
      }
      for (size_t i = 0; i < n; ++i) {
        if (!active[i]) continue;
        const int core = core_c;
        int ntiles = 0;  // of the widest block of the chunk
        for (int b = 0; b < chunk; ++b) {
          int L, R;
          h_rng_block(tj[i].pl, tj[i].tl, tj[i].sub, tj[i].s0 + b * T, tj[i].s0 + (b + 1) * T, &L, &R);
          if (R >= L) ntiles = std::max(ntiles, (R - L + core) / core);
        }
        // the tiles of jobs on packed sequences first (wfa_tile2_kernel), the others (an N, soft-masked bases: the byte kernel) behind them
        for (int d = 0; d < 2; ++d)
          for (int t = 0; t < ntiles; ++t) (tj[i].packed ? tasks : tasks_by).push_back(TileTask{(int32_t)i, d, t, core});  // (tile index, tile width): the kernel places it
      }
      const size_t n_pk = tasks.size();
      tasks.insert(tasks.end(), tasks_by.begin(), tasks_by.end());
      if (tasks.empty()) { h->err = "tile phase: active jobs without a tile"; return WFM_E_HIP; }
      if (h->tiletasks.ensure(tasks.size())) { h->err = "out of device memory (tile tasks)"; return WFM_E_NOMEM; }
      HIPCHK(h, hipMemcpyAsync(h->tiletasks.p, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice, h->stream));
      const auto tq1 = clk();
      // which instantiations of the packed kernel a block of the chunk can have tiles for (wfa_tile2_kernel, FINE): the one without per-score
      // maxima always (the blocks before a job's meeting block and the run up to the meeting point); the one with them where a job that simply
      // moved on has reached its fine_s, and in the chunk's first block for the jobs the last chunk left in mode 5
      const bool coarse_on = cfg.reg && cfg.exact && tile2_coarse_maxima();
      std::vector<int> variants_b((size_t)chunk, coarse_on ? 0 : 2);
      if (coarse_on) {
        for (size_t i = 0; i < n; ++i) {
          if (!active[i] || !(tj[i].packed & 1)) continue;
          for (int b = 0; b < chunk; ++b) {
            if ((tj[i].mode == 5 && b == 0) || (tj[i].mode == 0 && (int64_t)tj[i].s0 + (int64_t)(b + 1) * T >= (int64_t)tj[i].fine_s)) variants_b[(size_t)b] |= 2;
            // without maxima: a job that simply moved on and is still below its fine_s (it may also have met meanwhile: its run up to the meeting
            // point needs no maxima either -- and is taken by the FINE instantiation where that one is launched alone)
            if (tj[i].mode == 0 && (int64_t)tj[i].s0 + (int64_t)(b + 1) * T < (int64_t)tj[i].fine_s) variants_b[(size_t)b] |= 1;
          }
        }
        for (int b = 0; b < chunk; ++b) if (!variants_b[(size_t)b]) variants_b[(size_t)b] = 2;  // (only runs up to a meeting point: either would do)
      }
      for (int b = 0; b < chunk; ++b) {
        HIPCHK(h, hipEventRecord(h->tile_ev[2 * b], h->stream));
        if (cfg.reg) {
          if (n_pk) launch_tile2(S->d_pk, h->ring.p, h->tilejobs.p, h->tiletasks.p, h->tilemak.p, (int)n_pk, threads_b[(size_t)b], T, variants_b[(size_t)b], h->stream);
          if (tasks.size() > n_pk)
            launch_tile_reg(S->d_seq, h->ring.p, h->tilejobs.p, h->tiletasks.p + n_pk, h->tilemak.p, (int)(tasks.size() - n_pk), threads_b[(size_t)b], T, cfg.C, any_cut, h->stream);
        } else launch_tile(S->d_seq, h->ring.p, h->tilejobs.p, h->tiletasks.p, h->tilemak.p, (int)tasks.size(), cfg.threads, T, cfg.Wt, lds, dp, scope, ring_rows_for(scope), h->stream);
        HIPCHK(h, hipEventRecord(h->tile_ev[2 * b + 1], h->stream));
        launch_tile_advance(h->tilejobs.p, h->tilemak.p, (int)n, T, dp, (cfg.reg && cfg.exact) ? 1 : 0, coarse_on ? 1 : 0, variants_b[(size_t)b], h->stream);
      }
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(got.data(), h->tilejobs.p, n * sizeof(TileJob), hipMemcpyDeviceToHost, h->stream));
      const auto tq2 = clk();
      HIPCHK(h, hipStreamSynchronize(h->stream));
      const auto tq3 = clk();
      for (int b = 0; b < chunk; ++b) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, h->tile_ev[2 * b], h->tile_ev[2 * b + 1]));
        tile_ms += ms;
        if (h->call_base) {
          float t0 = 0;
          HIPCHK(h, hipEventElapsedTime(&t0, h->call_base, h->tile_ev[2 * b]));
          h->tile_iv.emplace_back(t0, t0 + ms);
        }
      }
      blocks += (uint32_t)chunk;
      for (int b = 0; b < chunk; ++b) lane_cells += (double)tasks.size() * threads_b[(size_t)b] * cfg.C * T;
      h->stats.tile_launches += (uint32_t)chunk;
      h->stats.tile_tasks += (uint32_t)(tasks.size() * (size_t)chunk);
      n_active = 0;
      for (size_t i = 0; i < n; ++i) {
        // cells of the blocks this job ran since the last look (the block that found the meeting point included)
        for (int bl = tj[i].nblocks; bl < got[i].nblocks; ++bl) {
          const bool last_exact = got[i].mode == 2 && bl == got[i].nblocks - 1;  // the block that stopped at the meeting point
          const int base = s_begin[i] + (last_exact ? bl - 1 : bl) * T;          // it re-ran the block before it
          for (int d = 0; d < 2; ++d) {
            const int steps = last_exact ? (d == 0 ? got[i].tf : got[i].tr) : T;
            tile_cells += (uint64_t)h_cells_sum(tj[i].pl, tj[i].tl, tj[i].sub, base + 1, base + steps);
          }
        }
        if (got[i].fine_s == -1 && tj[i].fine_s != -1)  // the block in which the directions met ran once more, for its per-score maxima (TileJob::fine_s)
          for (int d = 0; d < 2; ++d) tile_cells += (uint64_t)h_cells_sum(tj[i].pl, tj[i].tl, tj[i].sub, got[i].s0 + 1, got[i].s0 + T);
        if (got[i].reran > tj[i].reran)  // the block before the meeting block ran once more, for its gap rows (TileJob::ring_prev)
          for (int d = 0; d < 2; ++d) tile_cells += (uint64_t)h_cells_sum(tj[i].pl, tj[i].tl, tj[i].sub, got[i].s0 - T + 1, got[i].s0);
        tj[i] = got[i];
        active[i] = (char)(got[i].active != 0);
        fmax[i] = got[i].fmax; rmax[i] = got[i].rmax;
        n_active += active[i];
      }
      ms_prep += msd(tq0, tq1); ms_launch += msd(tq1, tq2); ms_wait += msd(tq2, tq3); ms_post += msd(tq3, clk());
    }
    if (getenv("WFM_DEBUG") && atoi(getenv("WFM_DEBUG")) > 1)
      fprintf(stderr, "[wfm] level %u tile phase, host side: task lists %.2f ms, launches %.2f ms, waiting for the device %.2f ms, bookkeeping %.2f ms\n", level, ms_prep, ms_launch, ms_wait, ms_post);
  }
  // cells that went into the result: both directions up to where the tile phase leaves the job (the full block in which
  // the wavefronts met was computed as well, and then again up to the meeting point: tile_cells counts it, this does not)
  uint64_t unique_level = 0;
  for (size_t i = 0; i < n; ++i) {
    const int sf_end = tj[i].mode == 2 ? tj[i].s0 + tj[i].tf : tj[i].s0, sr_end = tj[i].mode == 2 ? tj[i].s0 + tj[i].tr : tj[i].s0;
    unique_level += (uint64_t)h_cells_sum(tj[i].pl, tj[i].tl, tj[i].sub, s_begin[i] + 1, sf_end) +
                    (uint64_t)h_cells_sum(tj[i].pl, tj[i].tl, tj[i].sub, s_begin[i] + 1, sr_end);
  }
  h->stats.cells_tile_unique += unique_level;
  for (size_t i = 0; i < n; ++i) {
    BpJob& j = jobs[(size_t)tiled[i]];
    j.ring_off = tj[i].ring_in;
    ring2[i] = tj[i].ring_out;
    j.resume_s = tj[i].s0;
    j.resume_sr = -1; j.last_fwd = 0;
    if (tj[i].mode == 3) { j.resume_s = -3; continue; }  // ran out of its band: wfa_bp_kernel reports WFM_DEV_BAND
    if (tj[i].mode == 2) {  // stopped exactly at the meeting point: the step kernel goes straight to phase 2
      j.resume_s = tj[i].s0 + tj[i].tf;
      j.resume_sr = tj[i].s0 + tj[i].tr;
      j.last_fwd = tj[i].last_fwd;
    }
    j.fmax0 = fmax[i]; j.rmax0 = rmax[i];
  }
  if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] level %u: tiled %zu jobs, %u blocks of %d scores (Wt %d), %.3f ms; %.3e cells computed on %.3e lane-steps (%.0f %% of the lanes hold a cell), %.3e of them in the result (the block in which a job's wavefronts meet runs twice)\n", level, n, blocks, T, cfg.Wt, tile_ms,
                                   (double)tile_cells - cells_before, lane_cells, lane_cells > 0 ? 100.0 * ((double)tile_cells - cells_before) / lane_cells : 0.0, (double)unique_level);
  return WFM_OK;
}

// Phase 2 (overlap detection) of the jobs whose tile phase stopped exactly at the meeting point, without the
// step-by-step kernel: P2K more rows of both directions by the tile kernel, all rows kept (P2Job in wfa_device.h), then
// the reference's loop with only the tests left in it (wfa_p2_overlap_kernel).  cand: indices into jobs; res[q] is filled for every
// candidate, with status WFM_DEV_P2_MORE where the loop had not ended after 2 * P2K tests (wfa_bp_kernel takes those).
// A job whose walk ran out of rows goes another round when `may_continue`: its last RING rows of both directions are copied
// from the P2 rows into its ring (wfa_p2_to_ring_kernel) -- the state the sequential loop is in after 2 * P2K tests -- the
// breakpoint found so far is kept in carry[] and handed on as BpJob::best0, and the job's position in cand is appended to
// `again`.  (On divergent records a few jobs per level take 120 - 230 tests from the first touch of the wavefronts to a shared
// diagonal; finishing them step by step in wfa_bp_kernel was half of C2's device time.)
int run_p2_phase(wfm_handle* h, wfm_seqset* S, const DevPen& dp, int scope, const TileCfg& cfg, std::vector<BpJob>& jobs,
                 const std::vector<int>& cand, const std::vector<int64_t>& ring_other, std::vector<BpResult>& res, double& ms_out,
                 std::vector<BpResult>& carry, std::vector<char>& has_carry, bool may_continue, std::vector<int>& again) {
  if (cand.empty()) return WFM_OK;
  const int core = cfg.Wt - 2 * P2K;
  size_t i0 = 0;
  std::vector<TileJob> tj;
  std::vector<P2Job> pj;
  std::vector<TileTask> tasks, tasks_by;
  std::vector<BpResult> got;
  // the rows of a chunk of jobs may take a quarter of the budget (the rings hold the rest)
  const size_t budget = std::max<size_t>(h->mem_budget / 4, (size_t)64 << 20);
  while (i0 < cand.size()) {
    tj.clear(); pj.clear(); tasks.clear(); tasks_by.clear();
    size_t elems = 0, i = i0, maxw2 = 0, bm_elems = 0;
    for (; i < cand.size(); ++i) {
      const BpJob& j = jobs[(size_t)cand[i]];
      const int reach = std::max(j.resume_s, j.resume_sr) + P2K;
      int L, R;  // every diagonal a row of the window can hold: the snapshot's rows (26 back) and the rows computed ahead
      h_rng_block(j.pl, j.tl, j.sub, std::max(0, std::min(j.resume_s, j.resume_sr) - 27), reach, &L, &R);
      if (R < L) { L = 0; R = 0; }
      const int koff2 = ((-L + 4) + 3) & ~3;                       // column of diagonal 0: a multiple of 4, >= 4 columns of margin
      const size_t w2 = ((size_t)(R + koff2 + 8) + 3) & ~(size_t)3;
      const size_t nblk = (w2 >> 6) + 1;
      const size_t need = w2 * 2 * 5 * P2K, need_bm = nblk * 2 * P2ROWS * 5;
      if (!tj.empty() && (elems + need + 2 * (bm_elems + need_bm)) * 4 > budget) break;
      maxw2 = std::max(maxw2, w2);
      TileJob t{};
      t.p_fwd = j.p_fwd; t.t_fwd = j.t_fwd; t.p_rev = j.p_rev; t.t_rev = j.t_rev;
      t.ring_in = j.ring_off; t.ring_out = ring_other[i];
      t.pl = j.pl; t.tl = j.tl; t.comp_begin = j.comp_begin; t.comp_end = j.comp_end;
      t.width = j.width; t.koff = j.koff; t.s0 = 0; t.active = 1; t.fmax = 0; t.rmax = 0; t.nblocks = 0;
      t.mode = 4; t.tf = j.resume_s; t.tr = j.resume_sr; t.last_fwd = j.last_fwd; t.packed = j.packed;
      t.p2_off = (int64_t)elems; t.w2 = (int32_t)w2; t.koff2 = koff2; t.sub = j.sub;
      P2Job q{};
      q.ring_in = j.ring_off; q.p2_off = (int64_t)elems; q.width = j.width; q.koff = j.koff; q.w2 = (int32_t)w2; q.koff2 = koff2;
      q.pl = j.pl; q.tl = j.tl; q.sf = j.resume_s; q.sr = j.resume_sr; q.last_fwd = j.last_fwd; q.sub = j.sub; q.best0 = j.best0;
      q.nblk = (int32_t)nblk; q.bm_off = (int64_t)bm_elems;
      bm_elems += need_bm;
      elems += need;
      tj.push_back(t); pj.push_back(q);
    }
    const size_t n = tj.size();
    // one tile without a halo per job-direction while the widest range of the chunk fits one (see run_tiled_phase)
    int threads_c = cfg.threads, core_c = core;
    {
      int widest = 0;
      for (const TileJob& t : tj)
        for (int d = 0; d < 2; ++d) {
          int L, R;
          h_rng_block(t.pl, t.tl, t.sub, d == 0 ? t.tf : t.tr, (d == 0 ? t.tf : t.tr) + P2K, &L, &R);
          widest = std::max(widest, R - L + 1);
        }
      if (widest <= cfg.threads * 2) {
        threads_c = std::min(cfg.threads, std::max(64, ((widest + 1) / 2 + 63) / 64 * 64));
        core_c = threads_c * 2;
      }
    }
    for (size_t jn = 0; jn < n; ++jn)
      for (int d = 0; d < 2; ++d) {
        int Ld, Rd;
        h_rng_block(tj[jn].pl, tj[jn].tl, tj[jn].sub, d == 0 ? tj[jn].tf : tj[jn].tr, (d == 0 ? tj[jn].tf : tj[jn].tr) + P2K, &Ld, &Rd);
        const int ntiles = Rd >= Ld ? (Rd - Ld + core_c) / core_c : 0;
        for (int t2 = 0; t2 < ntiles; ++t2) (tj[jn].packed ? tasks : tasks_by).push_back(TileTask{(int32_t)jn, d, t2, core_c});
      }
    const size_t n_pk = tasks.size();  // (packed jobs' tiles first: see run_tiled_phase)
    tasks.insert(tasks.end(), tasks_by.begin(), tasks_by.end());
    if (h->p2rows.ensure(elems + 16) || h->p2max.ensure(n * 2 * P2ROWS * 5) || h->p2bmax.ensure(bm_elems + 16) || h->p2pbmax.ensure(bm_elems + 16) ||
        h->p2jobs.ensure(n) || h->tilejobs.ensure(n) || h->tiletasks.ensure(tasks.size()) || h->bpres.ensure(std::max(n, jobs.size()))) {
      h->err = "out of device memory (phase-2 rows)"; return WFM_E_NOMEM;
    }
    HIPCHK(h, hipMemcpyAsync(h->tilejobs.p, tj.data(), n * sizeof(TileJob), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->p2jobs.p, pj.data(), n * sizeof(P2Job), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->tiletasks.p, tasks.data(), tasks.size() * sizeof(TileTask), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    bool any_cut = false;
    for (const TileJob& t : tj) any_cut |= t.sub != SUB_NONE;
    if (n_pk) launch_tile2_p2(S->d_pk, h->ring.p, h->tilejobs.p, h->tiletasks.p, (int)n_pk, threads_c, h->p2rows.p, h->stream);
    if (tasks.size() > n_pk) launch_tile_p2(S->d_seq, h->ring.p, h->tilejobs.p, h->tiletasks.p + n_pk, (int)(tasks.size() - n_pk), threads_c, h->p2rows.p, any_cut, h->stream);
    launch_p2_blockmax(h->ring.p, h->p2rows.p, h->p2jobs.p, h->p2bmax.p, h->p2max.p, (int)n, h->stream);
    static const int p2_threads = getenv("WFM_P2_THREADS") ? atoi(getenv("WFM_P2_THREADS")) : 0;
    launch_p2_overlap(h->ring.p, h->p2rows.p, h->p2jobs.p, h->p2max.p, h->p2bmax.p, h->p2pbmax.p, h->bpres.p, (int)n,
                      p2_threads > 0 ? p2_threads : 1024, (int)(maxw2 >> 6) + 1, dp, scope, h->stream);  // 16 waves over the 5 x P2G (test, component) scans of a round and their rows
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    got.resize(n);
    HIPCHK(h, hipMemcpyAsync(got.data(), h->bpres.p, n * sizeof(BpResult), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    ms_out += ms;
    if (h->call_base) {
      float t0 = 0;
      HIPCHK(h, hipEventElapsedTime(&t0, h->call_base, h->ev0));
      h->bp_iv.emplace_back(t0, t0 + ms);
    }
    if (const char* de = getenv("WFM_P2_DUMP")) {  // diagnosis: the row maxima of one job of the chunk
      const size_t q = std::min<size_t>((size_t)atoi(de), n - 1);
      std::vector<int32_t> rm((size_t)2 * P2ROWS * 5);
      HIPCHK(h, hipMemcpy(rm.data(), h->p2max.p + q * 2 * P2ROWS * 5, rm.size() * 4, hipMemcpyDeviceToHost));
      const P2Job& pjq = pj[q];
      fprintf(stderr, "[wfm] p2 dump job %zu: pl %d tl %d sf %d sr %d last_fwd %d sub %d w2 %d nblk %d -> status %d score %d (fwd %d rev %d comp %d k %d) rounds %d\n", q, pjq.pl, pjq.tl, pjq.sf,
              pjq.sr, pjq.last_fwd, pjq.sub, pjq.w2, pjq.nblk, got[q].status, got[q].score, got[q].score_fwd, got[q].score_rev, got[q].comp, got[q].k_fwd, got[q].pad_);
      for (int d = 0; d < 2; ++d)
        for (int r = 0; r < P2ROWS; r += 3) {
          const int32_t* m = rm.data() + ((size_t)d * P2ROWS + r) * 5;
          fprintf(stderr, "[wfm]   dir %d row %d (s = %d): max M %d I1 %d I2 %d D1 %d D2 %d\n", d, r, (d == 0 ? pjq.sf : pjq.sr) - 25 + r, m[0], m[1], m[2], m[3], m[4]);
        }
    }
    h->stats.p2_launches++;
    h->stats.p2_jobs += (uint32_t)n;
    if (getenv("WFM_DEBUG")) {
      int more = 0;
      double tk = 0, tc = 0, rd = 0; uint32_t tkmax = 0; int rdmax = 0;
      for (const BpResult& r : got) { more += r.status == WFM_DEV_P2_MORE; tk += r.ticks_p2; tc += r.ticks_p1; rd += r.pad_; tkmax = std::max(tkmax, r.ticks_p2); rdmax = std::max(rdmax, r.pad_); }
      fprintf(stderr, "[wfm] phase 2 from rows computed ahead: %zu jobs, widest %zu columns, %zu tiles of %d threads, %.3f ms, %d left to the step kernel; walk per job: %.1f rounds (max %d), %.0f us (max %.0f), of which cells stage %.0f us\n",
              n, maxw2, tasks.size(), threads_c, ms, more, rd / n, rdmax, tk / n / 100.0, tkmax / 100.0, tc / n / 100.0);
      size_t qs = 0;
      for (size_t q = 0; q < n; ++q) if (got[q].ticks_p2 > got[qs].ticks_p2) qs = q;
      fprintf(stderr, "[wfm]   slowest walk: pl %d tl %d sub %d w2 %d nblk %d: %d rounds, %.0f us = listing %.0f + cells %.0f (incl. listing) + pick %.0f, %u blocks listed, status %d\n", pj[qs].pl, pj[qs].tl,
              pj[qs].sub == SUB_NONE ? -1 : pj[qs].sub, pj[qs].w2, pj[qs].nblk, got[qs].pad_, got[qs].ticks_p2 / 100.0, got[qs].ticks_list / 100.0, got[qs].ticks_p1 / 100.0, got[qs].ticks_pick / 100.0,
              got[qs].work_items, got[qs].status);
    }
    // another round for the jobs whose walk ran out of rows (while their rings have room for its rows)
    std::vector<P2Job> mj;
    for (size_t q = 0; q < n; ++q) {
      const size_t jq = (size_t)cand[i0 + q];
      BpJob& j = jobs[jq];
      if (got[q].status == WFM_DEV_P2_NOTHING) { got[q] = carry[jq]; continue; }  // nothing better than what an earlier round found
      if (got[q].status != WFM_DEV_P2_MORE) continue;
      if (got[q].comp >= 0) {  // a breakpoint so far (better than the one handed in, if any)
        carry[jq] = got[q]; carry[jq].status = 0; has_carry[jq] = 1;
        j.best0 = got[q].score;
      }
      if (!may_continue || (j.band > 0 && std::max(j.resume_s, j.resume_sr) + 2 * P2K + 2 > j.band)) continue;  // wfa_bp_kernel goes on from here
      mj.push_back(pj[q]);
      j.resume_s += P2K; j.resume_sr += P2K;  // 2 * P2K tests: both directions P2K rows further, the same one stepped last
      again.push_back((int)(i0 + q));
    }
    if (!mj.empty()) {
      HIPCHK(h, hipMemcpyAsync(h->p2jobs.p, mj.data(), mj.size() * sizeof(P2Job), hipMemcpyHostToDevice, h->stream));
      launch_p2_to_ring(h->ring.p, h->p2rows.p, h->p2jobs.p, (int)mj.size(), h->stream);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipStreamSynchronize(h->stream));  // (mj is read by the copy above)
    }
    for (size_t q = 0; q < n; ++q) res[(size_t)cand[i0 + q]] = got[q];
    i0 = i;
  }
  return WFM_OK;
}

// Aligns problems [first, last) of S; their op strings go to ops_arena from byte arena_base on.
// runs_out != nullptr: run-length output (wfm_align_batch_rle) -- the part's merged runs are appended to *runs_out and
// ops_off counts from the part's first run (the caller shifts the parts into one buffer); ops_arena is not touched
int align_resident_impl(wfm_handle* h, const wfm_penalties_t* pen, wfm_seqset* S, size_t first, size_t last, wfm_result_t* out,
                        char* ops_arena, size_t arena_bytes, size_t arena_base, std::vector<uint32_t>* runs_out = nullptr,
                        uint32_t* pflags = nullptr) {
  int scope = 0;
  int rc = validate_pen(pen, &scope);
  if (rc != WFM_OK) { h->err = "unsupported penalties"; return rc; }
  const auto t_start = std::chrono::steady_clock::now();
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = last - first;
  h->stats = wfm_stats_t{};
  if (n == 0) return 0;
  const DevPen dp{pen->x, pen->o1, pen->e1, pen->o2, pen->e2};

  // (read per call, not once per process: the tests run the bounded and the unbounded forms in one process)
  const bool use_hints = !(getenv("WFM_SCORE_HINT") && atoi(getenv("WFM_SCORE_HINT")) == 0);
  const bool use_bound = use_hints && !(getenv("WFM_BOUND") && atoi(getenv("WFM_BOUND")) == 0);
  const int slack_env = getenv("WFM_SUB_SLACK") ? atoi(getenv("WFM_SUB_SLACK")) : -1;  // tests: < 0 default, >= 2^28 none
  std::vector<int32_t> prob_status(S->meta.size(), WFM_ST_OK);  // indexed by problem id
  std::vector<uint64_t> prob_cells(S->meta.size(), 0);

  // RLE slot buffer (zero = empty)
  if (h->rle.ensure((size_t)S->rle_total + 16) || h->rle_out.ensure((size_t)S->rle_total + 16)) {
    h->err = "out of device memory (rle)"; return WFM_E_NOMEM;
  }
  HIPCHK(h, hipMemsetAsync(h->rle.p, 0, ((size_t)S->rle_total + 16) * sizeof(uint32_t), h->stream));

  // roots
  std::vector<Node> bp_nodes, base_nodes, next_bp, retry;
  for (size_t i = first; i < last; ++i) {
    const ProbMeta& pm = S->meta[i];
    Node nd{};
    nd.prob = (int32_t)i; nd.pb = 0; nd.pl = pm.plen; nd.tb = 0; nd.tl = pm.tlen;
    nd.cb = C_M; nd.ce = C_M; nd.score_rem = INT_MAX; nd.endsfree = 0;
    nd.sub = SUB_NONE; nd.hinted = 0;
    if (pflags && pm.mode == WFM_MODE_END2END_BIWFA && !((size_t)i < S->acgt.size() && S->acgt[i])) pflags[i] |= WFM_PF_BYTE_KERNEL;
    if (use_hints && pm.hint > 0 && pm.mode == WFM_MODE_END2END_BIWFA) { nd.sub = pm.hint; nd.hinted = 1; }
    const int64_t bound = (int64_t)gapcost(*pen, pm.plen) + gapcost(*pen, pm.tlen) + 8;
    if (pm.mode == WFM_MODE_ENDSFREE) {
      nd.endsfree = 1;
      // ends-free score is bounded by the cheaper all-gap alignment; start small, double on overflow
      nd.smax = (int32_t)std::min<int64_t>(bound, 256);
      base_nodes.push_back(nd);
    } else if (pm.mode == WFM_MODE_END2END_UNI || std::max(pm.plen, pm.tlen) <= BIALIGN_FALLBACK_MIN_LENGTH ||
               pm.plen == 0 || pm.tlen == 0) {
      nd.smax = (int32_t)std::min<int64_t>(bound, 256);
      base_nodes.push_back(nd);
    } else {
      bp_nodes.push_back(nd);
    }
  }

  // ---- an upper bound of every long root's score, from one greedy walk per root (wfa_bound_kernel): rigorous, so the root's
  // wavefronts are cut to what an alignment of at most that score can touch -- usually a fifth of what the caller's guess
  // leaves.  The guess stays where the walk gives up (divergent records, structural differences).
  uint64_t bounded_roots = 0, bound_gain = 0;
  double bound_ms = 0;
  {
    std::vector<BoundJob> bj;
    std::vector<size_t> owner;
    if (use_bound)
      for (size_t q = 0; q < bp_nodes.size(); ++q) {
        const Node& nd = bp_nodes[q];
        const ProbMeta& pm = S->meta[nd.prob];
        // only where a bound can bind: the end diagonal far from the start diagonal (see BpJob::sub below)
        if (pm.mode != WFM_MODE_END2END_BIWFA || std::min(nd.pl, nd.tl) < 1024 || std::abs(nd.tl - nd.pl) < 64) continue;
        bj.push_back(BoundJob{pm.p_fwd, pm.t_fwd, nd.pl, nd.tl});
        owner.push_back(q);
      }
    if (!bj.empty()) {
      const auto tb0 = std::chrono::steady_clock::now();
      if (h->bndjobs.ensure(bj.size()) || h->bndres.ensure(bj.size())) { h->err = "out of device memory (score bounds)"; return WFM_E_NOMEM; }
      HIPCHK(h, hipMemcpyAsync(h->bndjobs.p, bj.data(), bj.size() * sizeof(BoundJob), hipMemcpyHostToDevice, h->stream));
      launch_bound(S->d_seq, h->bndjobs.p, h->bndres.p, (int)bj.size(), dp, h->stream);
      HIPCHK(h, hipGetLastError());
      std::vector<int32_t> ub(bj.size());
      HIPCHK(h, hipMemcpyAsync(ub.data(), h->bndres.p, bj.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (size_t q = 0; q < bj.size(); ++q) {
        if (ub[q] < 0) continue;
        Node& nd = bp_nodes[owner[q]];
        if (nd.sub == SUB_NONE || ub[q] < nd.sub) {
          if (nd.sub != SUB_NONE) bound_gain += (uint64_t)(nd.sub - ub[q]);
          nd.sub = ub[q];
          nd.hinted = 1;  // (cannot fail; the retry of a root that runs past its bound stays as the safety net it is)
          ++bounded_roots;
        }
      }
      bound_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
      if (getenv("WFM_DEBUG"))
        fprintf(stderr, "[wfm] score bounds: %zu roots walked in %.2f ms, %llu bounded (on average %.0f below the caller's guess)\n", bj.size(), bound_ms,
                (unsigned long long)bounded_roots, bounded_roots ? (double)bound_gain / (double)bounded_roots : 0.0);
    }
  }

  LevelTimer tm;
  double wall_tile = 0, wall_base = 0;
  std::vector<BpJob> jobs;
  std::vector<int> tiled;
  std::vector<int64_t> ring2, ring3;
  std::vector<size_t> ring_third;  // elements of one ring of every tiled job of the chunk
  static const bool ring3_on = !(getenv("WFM_TILE_RING3") && atoi(getenv("WFM_TILE_RING3")) == 0);
  std::vector<int32_t> fine_from;
  const int fine_margin = getenv("WFM_TILE_FINE_MARGIN") ? atoi(getenv("WFM_TILE_FINE_MARGIN")) : 48;
  const int coarse_min_blocks = getenv("WFM_TILE_COARSE_MIN_BLOCKS") ? atoi(getenv("WFM_TILE_COARSE_MIN_BLOCKS")) : 32;
  const int coarse_max_jobs = getenv("WFM_TILE_COARSE_MAX_JOBS") ? atoi(getenv("WFM_TILE_COARSE_MAX_JOBS")) : 128;
  const TileCfg tcfg = tile_cfg(*pen, scope);
  // WFM_TILE_V2=0: every tile on the byte kernel (wfa_tile_reg_kernel) -- the A/B switch of the packed kernel (wfa_tile2.hip)
  const bool tile_v2 = !(getenv("WFM_TILE_V2") && atoi(getenv("WFM_TILE_V2")) == 0);
  const int RR = ring_rows_for(scope);  // rows of every ring of this call
  uint64_t tile_cells_level = 0;
  uint64_t band_retries = 0, band_jobs = 0, roots_banded = 0, roots_out = 0, hint_retries = 0, hinted_roots = 0;
  bool roots_off = false;
  std::vector<int32_t> node_of;
  std::vector<BpResult> res;
  uint32_t level = 0;
  while (!bp_nodes.empty() || !base_nodes.empty()) {
    ++level;
    // ---- breakpoint jobs of this level (chunked to the memory budget) ----
    next_bp.clear();
    // Rings cover every diagonal of a job ((pl + tl) columns of 1280 B, twice for tiled jobs): fine for a batch of
    // few deep problems, wasteful for thousands of long low-divergence records, whose wavefronts stay within a few
    // thousand diagonals and which would otherwise be worked off in many small chunks.  When the level does not fit
    // the budget, jobs get rings for |k| <= band only: a child's total score is known (half of it per direction,
    // plus the overlap phase), a root gets WFM_BAND_ROOT scores; whoever runs out of its band is run again on a
    // full ring.  WFM_BAND=0 switches this off.
    bool use_band = false, over_budget = false;
    {
      const char* be = getenv("WFM_BAND");
      const int band_on = be ? atoi(be) : 1;
      size_t total = 0;
      for (const Node& nd : bp_nodes) total += (((size_t)nd.pl + nd.tl + 9 + 3) & ~(size_t)3) * 2 * 5 * RR * 2;
      // (narrow rings whenever full ones would take more than 2 GB, not only when they would not fit: every fresh GB of a
      // first hipMalloc costs ~30 ms on this driver, scripts/micro/malloc_cost2.hip, and a one-shot run pays it)
      use_band = band_on && total * 4 > std::min<size_t>(h->mem_budget, (size_t)2 << 30);
      over_budget = total * 4 > h->mem_budget;
    }
    static const size_t ring_chunk_bytes = (size_t)(getenv("WFM_RING_CHUNK_GB") ? std::max(1, atoi(getenv("WFM_RING_CHUNK_GB"))) : 4) << 30;
    const char* bre = getenv("WFM_BAND_ROOT");
    const int band_root = bre ? std::max(64, atoi(bre)) : 4096;
    size_t i0 = 0;
    while (i0 < bp_nodes.size()) {
      jobs.clear();
      node_of.clear();
      size_t ring_elems = 0;
      size_t i = i0;
      int maxw = 0;
      tiled.clear(); ring2.clear(); ring3.clear(); ring_third.clear(); fine_from.clear();
      int64_t fine_min_blocks = INT64_MAX;  // fewest blocks any tiled job of the chunk is expected to run before its directions meet
      for (; i < bp_nodes.size(); ++i) {
        const Node& nd = bp_nodes[i];
        const ProbMeta& pm = S->meta[nd.prob];
        size_t width = ((size_t)nd.pl + nd.tl + 9 + 3) & ~(size_t)3;  // columns 4 .. pl+tl+4, 16-byte chunks
        int koff = nd.pl + 4;
        bool tile_it = tcfg.enabled && nd.pl + nd.tl >= tcfg.min_len &&
                             (nd.score_rem == INT_MAX || nd.score_rem >= tcfg.min_score);
        int band = 0;
        // (a root without a bound gets a guessed band only when the level would not fit otherwise: below the budget the
        // guess has nothing to win and a deep record -- 5 % divergence: 6 k scores per direction -- everything to lose)
        const bool known = nd.score_rem != INT_MAX || nd.sub != SUB_NONE;
        if (use_band && (known || over_budget) && tile_it && !nd.noband && !(roots_off && nd.score_rem == INT_MAX)) {
          // scores one direction is allowed to reach; the ring holds |k| <= band + 8, its left margin stays 4 columns
          // (a root under a bound of its score leaves the tile phase once a direction passes (bound + 128) / 2)
          int64_t dir_scores = nd.score_rem == INT_MAX ? (int64_t)band_root : (int64_t)nd.score_rem / 2 + 64;
          if (nd.score_rem == INT_MAX && nd.sub != SUB_NONE) dir_scores = std::min<int64_t>(dir_scores, ((int64_t)nd.sub + 128) / 2 + 64);
          const int64_t b = dir_scores + (int64_t)tcfg.chunk * tcfg.T + 16;
          const int64_t shift = ((int64_t)nd.pl - (b + 8)) & ~(int64_t)3;  // columns cut off on the left, whole 16-byte chunks
          const int64_t right = std::min<int64_t>(nd.tl, b + 8);           // largest diagonal kept
          if (shift > 0) {
            const size_t w = ((size_t)((int64_t)nd.pl - shift + right + 9) + 3) & ~(size_t)3;
            if (w * 2 <= width) { band = (int)b; width = w; koff = (int)(nd.pl + 4 - shift); }
          }
        }
        if (tile_it && width * 2 * 5 * RR * 2 * 4 > h->mem_budget) tile_it = false;  // two snapshot rings do not fit: step-by-step kernel
        const size_t need = width * 2 * 5 * RR * (tile_it ? 2 : 1);
        // (a chunk of a level stops at 4 GB of rings even when the budget allows more: hundreds of jobs fill the device
        // long before that, and every GB of a first allocation costs 30 - 70 ms.  C1 substitute, three handles in a fresh
        // process: 8 GB chunks 8.3 s cold / 5.33 s warm, 4 GB 5.67 / 5.52, 2 GB 6.22 / 6.06 -- scripts/c1_cold.sh.  A chunk of
        // fewer than 128 jobs may grow to 8 GB: C3's 21 roots of a part are 5.4 GB of full rings, and cut in two they fill the device worse.
        // Tried and dropped: two launches per block, jobs without a score bound apart from those with one -- the plain kernel form
        // has 7 % fewer instructions, the second launch cost more: C2 0.18 -> 0.21 s, C1 no better)
        if (!jobs.empty() && (ring_elems + need) * 4 > std::min<size_t>(h->mem_budget, jobs.size() >= 128 ? ring_chunk_bytes : std::max(ring_chunk_bytes, (size_t)8 << 30))) break;
        if (need * 4 > h->mem_budget) { prob_status[nd.prob] = WFM_ST_OOM; continue; }
        BpJob j{};
        j.p_fwd = pm.p_fwd + nd.pb;
        j.t_fwd = pm.t_fwd + nd.tb;
        j.p_rev = pm.p_rev + (pm.plen - nd.pb - nd.pl);
        j.t_rev = pm.t_rev + (pm.tlen - nd.tb - nd.tl);
        j.ring_off = (int64_t)ring_elems;
        j.pl = nd.pl; j.tl = nd.tl;
        j.comp_begin = nd.cb; j.comp_end = nd.ce;
        j.width = (int32_t)width;
        j.koff = koff;
        j.resume_s = -1; j.resume_sr = -1; j.last_fwd = 0; j.fmax0 = 0; j.rmax0 = 0;
        j.band = band;
        // a bound only earns its keep when the end diagonal is far from the start diagonal relative to the score (padded
        // records and their children): for a balanced problem it starts to bind where the wavefronts meet, and costs the
        // tile kernel its bookkeeping all the way there
        j.sub = (nd.sub != SUB_NONE && (int64_t)std::abs(nd.tl - nd.pl) * 8 >= (int64_t)nd.sub) ? nd.sub : SUB_NONE;
        j.best0 = 0;
        j.packed = (tile_v2 && tcfg.reg && tcfg.C == 2 && (size_t)nd.prob < S->acgt.size() && S->acgt[(size_t)nd.prob]) ? 1 : 0;
        // bit 1: near-identical sequences -- the job's score is known (a child's, a bounded or hinted root's) to be under a sixteenth of its length; the packed
        // tile kernel then hands a lone long run to the whole wave at once (wfa_tile2.hip, tail_direct).  Whether the bound also CUTS the rows (sub below) is another matter.
        if (j.packed && nd.sub != SUB_NONE && (int64_t)nd.sub * 16 < (int64_t)nd.pl + nd.tl) j.packed |= 2;
        band_jobs += band > 0;
        if (tile_it) {
          tiled.push_back((int)jobs.size()); ring2.push_back((int64_t)(ring_elems + need / 2)); ring_third.push_back(need / 2);
          // per-score maxima from here on (TileJob::fine_s): a child's directions meet near half its score (the trigger -- the sum of the two largest
          // antidiagonals -- can fire a little earlier, never later); a root's score is anybody's guess: it finds its meeting block with one maximum
          // per block and runs it again (whether the chunk uses any of this is decided below, once its jobs are known)
          int ff;
          const int64_t est = nd.score_rem != INT_MAX ? (int64_t)nd.score_rem : (nd.sub != SUB_NONE ? (int64_t)nd.sub : (int64_t)nd.pl + nd.tl);  // its score / the guess or bound / the worst case
          fine_min_blocks = std::min<int64_t>(fine_min_blocks, est / 2 / tcfg.T);
          if (nd.score_rem != INT_MAX) ff = std::max(0, nd.score_rem / 2 - fine_margin);
          else ff = INT_MAX;
          fine_from.push_back(ff);
        }
        node_of.push_back((int32_t)i);
        ring_elems += need;
        // widest wavefront this job can reach: 2 diagonals per score of one direction (~half the total score)
        const int64_t est_w = nd.score_rem == INT_MAX ? (int64_t)width : std::min<int64_t>((int64_t)width, (int64_t)nd.score_rem + 128);
        maxw = std::max(maxw, (int)est_w);
        jobs.push_back(j);
      }
      const size_t chunk_end = i;
      // One maximum per block instead of one per score (TileJob::fine_s) pays where a launch is a few deep jobs that move in step -- C3: 21 roots
      // of 78 blocks each, -3.7 % per step -- and costs where it is hundreds of jobs of all depths: their blocks need both instantiations of the
      // kernel side by side (two launches per block), and a root runs its meeting block a third time behind one more look of the host: C2 +10 %
      // device time, the scaled C4 rank +3 % (gpurun_out/r6r/ab2.log).  So: only chunks of at most coarse_max_jobs jobs, every one of them
      // at least coarse_min_blocks blocks deep; everybody else keeps the per-score maxima from the first block on (one launch per block, as before).
      if (tiled.size() > (size_t)coarse_max_jobs || fine_min_blocks < (int64_t)coarse_min_blocks) std::fill(fine_from.begin(), fine_from.end(), 0);
      // third rings behind the chunk's rings (TileJob::ring_prev), for all of its tiled jobs or for none: where half as much again still fits the
      // budget (and 12 GB: fresh memory is 30 ms per GB).  The chunk's composition does not depend on it.
      {
        size_t third = 0;
        for (size_t x : ring_third) third += x;
        const bool give = ring3_on && tcfg.reg && tcfg.exact && third > 0 && (ring_elems + third) * 4 <= std::min<size_t>(h->mem_budget, (size_t)12 << 30);
        ring3.assign(tiled.size(), -1);
        if (give)
          for (size_t q = 0; q < tiled.size(); ++q) { ring3[q] = (int64_t)ring_elems; ring_elems += ring_third[q]; }
      }
      if (!jobs.empty()) {
        if (h->ring.ensure(ring_elems + 16) || h->bpjobs.ensure(jobs.size()) || h->bpres.ensure(jobs.size())) {
          h->err = "out of device memory (ring arena)"; return WFM_E_NOMEM;
        }
        {
          double tms = 0; uint64_t tcells = 0;
          const auto tw0 = std::chrono::steady_clock::now();
          rc = run_tiled_phase(h, S, dp, scope, tcfg, tcfg.T, false, jobs, tiled, ring2, tms, tcells, level, &fine_from, &ring3);
          if (rc == WFM_OK && tcfg.T_refine > 0 && tcfg.T_refine < tcfg.T && !(tcfg.reg && tcfg.exact))
            rc = run_tiled_phase(h, S, dp, scope, tcfg, tcfg.T_refine, true, jobs, tiled, ring2, tms, tcells, level);
          wall_tile += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
          if (rc != WFM_OK) return rc;
          tm.bp_ms += tms; tm.tile_ms += tms;
          h->stats.cells_bp += tcells; h->stats.cells_tile += tcells;
          for (size_t q = 0; q < tiled.size(); ++q) (void)q;
          tile_cells_level = tcells;
        }
        res.assign(jobs.size(), BpResult{});
        // ---- phase 2 of the jobs the tile phase left exactly at their meeting point: rows computed ahead + scan + replay
        std::vector<int> rest;  // jobs for the step kernel: not tiled, not exact, or not finished by the rows computed ahead
        std::vector<BpResult> carry;  // breakpoints found by rounds of phase 2 that did not end the walk
        std::vector<char> has_carry;
        std::vector<int> more_set;
        {
          static const bool p2_on = !(getenv("WFM_P2") && atoi(getenv("WFM_P2")) == 0);
          std::vector<int> cand;
          std::vector<int64_t> other;
          std::vector<char> is_cand(jobs.size(), 0);
          if (p2_on && tcfg.reg && tcfg.exact)
            for (size_t q = 0; q < tiled.size(); ++q) {
              const BpJob& j = jobs[(size_t)tiled[q]];
              if (j.resume_s >= 0 && j.resume_sr >= 0) { cand.push_back(tiled[q]); other.push_back(ring2[q]); is_cand[(size_t)tiled[q]] = 1; }
            }
          double pms = 0;
          carry.assign(jobs.size(), BpResult{});
          has_carry.assign(jobs.size(), 0);
          const int p2_rounds = getenv("WFM_P2_ROUNDS") ? std::max(1, atoi(getenv("WFM_P2_ROUNDS"))) : 64;
          std::vector<int> cand_r = cand, again;
          std::vector<int64_t> other_r = other;
          for (int round = 1; !cand_r.empty(); ++round) {
            again.clear();
            rc = run_p2_phase(h, S, dp, scope, tcfg, jobs, cand_r, other_r, res, pms, carry, has_carry, round < p2_rounds, again);
            if (rc != WFM_OK) return rc;
            std::vector<int> c2; std::vector<int64_t> o2;
            for (int a : again) {
              c2.push_back(cand_r[(size_t)a]); o2.push_back(other_r[(size_t)a]);
              if (pflags) pflags[bp_nodes[(size_t)node_of[(size_t)cand_r[(size_t)a]]].prob] |= WFM_PF_P2_ROUNDS;
            }
            h->stats.p2_again += (uint32_t)again.size();
            cand_r.swap(c2); other_r.swap(o2);
          }
          tm.bp_ms += pms;
          for (size_t q = 0; q < jobs.size(); ++q)
            if (!is_cand[q] || res[q].status == WFM_DEV_P2_MORE) { rest.push_back((int)q); h->stats.p2_more += is_cand[q]; if (is_cand[q]) more_set.push_back((int)q); }
          if (pflags)  // jobs whose overlap walk went past the first round of rows computed ahead (or was finished by the step kernel)
            for (size_t q = 0; q < jobs.size(); ++q)
              if (is_cand[q] && res[q].status == WFM_DEV_P2_MORE) pflags[bp_nodes[(size_t)node_of[q]].prob] |= WFM_PF_P2_ROUNDS;
        }
        auto is_more = [&](int q) { return std::find(more_set.begin(), more_set.end(), q) != more_set.end(); };
        if (!rest.empty()) {
          // workgroup size: wide wavefronts want all 16 waves of a CU
          int threads = 1024;
          if (maxw <= 1024) threads = 256;
          else if (maxw <= 8192) threads = 512;
          std::vector<BpJob> rj(rest.size());
          for (size_t q = 0; q < rest.size(); ++q) rj[q] = jobs[(size_t)rest[q]];
          HIPCHK(h, hipMemcpyAsync(h->bpjobs.p, rj.data(), rj.size() * sizeof(BpJob), hipMemcpyHostToDevice, h->stream));
          HIPCHK(h, hipEventRecord(h->ev0, h->stream));
          launch_bp(S->d_seq, h->ring.p, h->bpjobs.p, h->bpres.p, (int)rj.size(), threads, dp, scope, RR, h->stream);
          HIPCHK(h, hipGetLastError());
          HIPCHK(h, hipEventRecord(h->ev1, h->stream));
          std::vector<BpResult> rr(rj.size());
          HIPCHK(h, hipMemcpyAsync(rr.data(), h->bpres.p, rr.size() * sizeof(BpResult), hipMemcpyDeviceToHost, h->stream));
          HIPCHK(h, hipStreamSynchronize(h->stream));
          float ms = 0;
          HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
          tm.bp_ms += ms;
          if (h->call_base) {
            float t0 = 0;
            HIPCHK(h, hipEventElapsedTime(&t0, h->call_base, h->ev0));
            h->bp_iv.emplace_back(t0, t0 + ms);
          }
          h->stats.bp_launches++;
          for (size_t q = 0; q < rest.size(); ++q) {
            if (rr[q].status == WFM_DEV_P2_NOTHING) {  // (only jobs that carry a breakpoint are handed a best0)
              const uint64_t c = rr[q].cells; const int32_t st = rr[q].steps;
              rr[q] = carry[(size_t)rest[q]]; rr[q].cells = c; rr[q].steps = st;
            }
            res[(size_t)rest[q]] = rr[q];
          }
          if (getenv("WFM_DEBUG")) {
            uint64_t c = 0; double t1 = 0, t2 = 0; int64_t st1 = 0, st = 0; uint32_t m1 = 0, m2 = 0;
            for (const BpResult& r : rr) { c += r.cells; t1 += r.ticks_p1; t2 += r.ticks_p2; st1 += r.steps_p1; st += r.steps; m1 = std::max(m1, r.ticks_p1); m2 = std::max(m2, r.ticks_p2); }
            fprintf(stderr, "[wfm] level %u: %zu bp jobs (step kernel), %d thr, %.3f ms, cells %.3e, avg steps p1 %.0f p2 %.0f, avg ms p1 %.3f p2 %.3f, max ms p1 %.3f p2 %.3f\n", level, rr.size(), threads, ms,
                    (double)c, (double)st1 / rr.size(), (double)(st - st1) / rr.size(), t1 / rr.size() / 1e5, t2 / rr.size() / 1e5, m1 / 1e5, m2 / 1e5);
            if (atoi(getenv("WFM_DEBUG")) > 1) {  // the slowest three
              std::vector<size_t> ord(rr.size());
              for (size_t q = 0; q < ord.size(); ++q) ord[q] = q;
              std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return rr[a].ticks_p1 + rr[a].ticks_p2 > rr[b].ticks_p1 + rr[b].ticks_p2; });
              for (size_t q = 0; q < std::min<size_t>(3, ord.size()); ++q) {
                const BpResult& r = rr[ord[q]];
                const BpJob& j = rj[ord[q]];
                fprintf(stderr, "[wfm]   slow step-kernel job: pl %d tl %d width %d band %d sub %d resume %d/%d (%s), status %d score %d = %d + %d, steps p1 %d p2 %d, ms p1 %.3f p2 %.3f\n", j.pl, j.tl, j.width,
                        j.band, j.sub == SUB_NONE ? -1 : j.sub, j.resume_s, j.resume_sr, is_more(rest[ord[q]]) ? "phase-2 walk ran out of rows" : (j.resume_sr >= 0 ? "exact" : "not exact"), r.status,
                        r.score, r.score_fwd, r.score_rev, r.steps_p1, r.steps - r.steps_p1, r.ticks_p1 / 1e5, r.ticks_p2 / 1e5);
              }
            }
          }
        }
        h->stats.bp_jobs += (uint32_t)jobs.size();
        const int dbg_lvl = getenv("WFM_DEBUG") ? std::max(1, atoi(getenv("WFM_DEBUG"))) : 0;  // (once per chunk, not once per job)
        for (size_t q = 0; q < jobs.size(); ++q) {
          const Node nd = bp_nodes[(size_t)node_of[q]];  // a copy: retries are appended to bp_nodes below
          const BpResult& r = res[q];
          prob_cells[nd.prob] += r.cells;
          h->stats.cells_bp += r.cells;
          if (nd.score_rem == INT_MAX && jobs[q].band > 0) { ++roots_banded; roots_out += r.status == WFM_DEV_BAND; }
          const bool guessed = nd.hinted && jobs[q].sub != SUB_NONE;  // the job really ran under the caller's guess
          if (r.status == WFM_DEV_BAND || (guessed && (r.status < 0 || (r.status == 0 && r.score > nd.sub)))) {
            // ran out of its narrow ring, or past the caller's guess of its score: once more, at the end of this level, on
            // a full ring and without the guess
            Node again = nd; again.noband = 1; again.sub = SUB_NONE; again.hinted = 0;
            if (pflags) pflags[nd.prob] |= nd.score_rem == INT_MAX ? WFM_PF_ROOT_AGAIN : WFM_PF_JOB_AGAIN;
            // (it joins the next level's jobs instead of holding this level up on its own: nodes are independent, only the gather at
            // the end waits for all of them.  Until round 5 a job that ran out of its ring was run again at the end of its own level --
            // three chains of 30 - 40 tile blocks one after the other in the first level of an LPA batch, 19 of its 50 ms of tile time;
            // WFM_RETRY_SAME_LEVEL=1 restores that for A/B runs)
            static const bool same_level = getenv("WFM_RETRY_SAME_LEVEL") && atoi(getenv("WFM_RETRY_SAME_LEVEL")) != 0;
            if (guessed || !same_level) next_bp.push_back(again); else bp_nodes.push_back(again);
            ++band_retries;
            hint_retries += guessed;
            continue;
          }
          if (r.status == 1) {  // end reached at score 0 -> base aligner
            Node b = nd; b.smax = 0; base_nodes.push_back(b);
          } else if (r.status != 0) {
            if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] problem %d: bialign job pl %d tl %d cb %d ce %d score_rem %d status %d (steps %d)\n", nd.prob, nd.pl, nd.tl, nd.cb, nd.ce, nd.score_rem, r.status, r.steps);
            prob_status[nd.prob] = WFM_ST_UNREACHABLE;
          } else {
            const int bp_h = r.off_fwd, bp_v = r.off_fwd - r.k_fwd;
            if (bp_h < 0 || bp_v < 0 || bp_h > nd.tl || bp_v > nd.pl) {
              if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] problem %d: bialign job pl %d tl %d (at %d, %d of the problem; level %u, begin / end components %d %d, bound %d%s): breakpoint (%d, %d) outside, score %d = %d + %d comp %d k %d\n", nd.prob, nd.pl, nd.tl, nd.pb, nd.tb, level, nd.cb, nd.ce, jobs[q].sub, nd.hinted ? " guessed" : "", bp_v, bp_h, r.score, r.score_fwd, r.score_rev, r.comp, r.k_fwd);
              if (const char* dd = getenv("WFM_DUMP_FAIL")) {  // diagnosis: the whole problem's sequences, for a replay
                const ProbMeta& pm = S->meta[nd.prob];
                std::vector<char> pb((size_t)pm.plen), tb((size_t)pm.tlen);
                (void)hipMemcpy(pb.data(), S->d_seq + pm.p_fwd, pb.size(), hipMemcpyDeviceToHost);
                (void)hipMemcpy(tb.data(), S->d_seq + pm.t_fwd, tb.size(), hipMemcpyDeviceToHost);
                static std::atomic<int> nfail{0};
                const std::string fn = std::string(dd) + "/fail_" + std::to_string(nfail.fetch_add(1)) + ".txt";
                if (FILE* f = fopen(fn.c_str(), "w")) {
                  fprintf(f, "%d %d %d\n", pm.plen, pm.tlen, pm.hint);
                  fwrite(pb.data(), 1, pb.size(), f); fputc('\n', f);
                  fwrite(tb.data(), 1, tb.size(), f); fputc('\n', f);
                  fclose(f);
                }
              }
              prob_status[nd.prob] = WFM_ST_UNREACHABLE; continue;
            }
            if (dbg_lvl > 1) fprintf(stderr, "[wfm] problem %d level %u: job pl %d tl %d cb %d ce %d rem %d -> bp v %d h %d score %d = %d + %d comp %d\n", nd.prob, level, nd.pl, nd.tl, nd.cb, nd.ce, nd.score_rem, bp_v, bp_h, r.score, r.score_fwd, r.score_rev, r.comp);
            Node a{}, b{};
            a.prob = nd.prob; a.pb = nd.pb; a.pl = bp_v; a.tb = nd.tb; a.tl = bp_h;
            // what a child can cost: the score its parent found for it, plus the opening of a gap it begins or ends in
            // (counted on the other side of the breakpoint)
            const int slack = slack_env >= 0 ? slack_env : 2 * std::max(pen->o1, pen->o2) + 8;
            a.sub = (int)std::min<int64_t>((int64_t)r.score_fwd + slack, SUB_NONE);
            b.sub = (int)std::min<int64_t>((int64_t)r.score_rev + slack, SUB_NONE);
            a.hinted = 0; b.hinted = 0;
            a.cb = nd.cb; a.ce = r.comp; a.score_rem = r.score_fwd;
            b.prob = nd.prob; b.pb = nd.pb + bp_v; b.pl = nd.pl - bp_v; b.tb = nd.tb + bp_h; b.tl = nd.tl - bp_h;
            b.cb = r.comp; b.ce = nd.ce; b.score_rem = r.score_rev;
            for (Node* c : {&a, &b}) {
              if (c->pl == 0 || c->tl == 0) { c->smax = 0; base_nodes.push_back(*c); }
              else if (c->score_rem <= BIALIGN_FALLBACK_MIN_SCORE) {
                // the leaf's own forward score: what the breakpoint credited it with, plus the opening of a gap it has
                // to end in (counted on the other side of that breakpoint)
                const int open_end = c->ce == C_M ? 0 : ((c->ce == C_I1 || c->ce == C_D1) ? pen->o1 : pen->o2);
                c->smax = std::max(c->score_rem, 0) + open_end;
                base_nodes.push_back(*c);
              }
              else next_bp.push_back(*c);
            }
          }
        }
      }
      // a root's band is a guess (its score is not known): when the guess keeps failing -- a batch of divergent
      // records -- the remaining roots get full rings right away instead of paying for the attempt
      if (!roots_off && roots_banded >= 16 && roots_out * 4 > roots_banded) roots_off = true;
      i0 = chunk_end;
    }
    bp_nodes.swap(next_bp);
    // ---- base jobs collected so far (incl. retries with a larger budget) ----
    // Leaves do not feed the recursion: while bialign jobs are left they wait (round 5), and all levels' leaves go out together behind the
    // last level -- two launches of thousands of leaves instead of two of hundreds per level, each with its wait for the device in the chain of
    // the batch's launches (C2: 12 launches + waits per part -> 3).  WFM_LEAVES_PER_LEVEL=1: the round-4 order; a quarter of a million leaves
    // waiting are run anyway (their arenas are chunked to the budget either way).
    static const bool leaves_per_level = getenv("WFM_LEAVES_PER_LEVEL") && atoi(getenv("WFM_LEAVES_PER_LEVEL")) != 0;
    if (!leaves_per_level && !bp_nodes.empty() && base_nodes.size() < ((size_t)1 << 18)) continue;
    while (!base_nodes.empty()) {
      retry.clear();
      const auto tb0 = std::chrono::steady_clock::now();
      rc = run_base_jobs(h, S, *pen, base_nodes, retry, prob_status, prob_cells, tm, pflags);
      wall_base += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
      if (rc != WFM_OK) return rc;
      base_nodes.swap(retry);
    }
  }
  h->stats.levels = level;
  if (getenv("WFM_P2_COUNT") && atoi(getenv("WFM_P2_COUNT"))) {
    unsigned long long c[8];
    wfm::p2_counters(c);
    fprintf(stderr, "[wfm] p2 overlap (cumulative): tests %llu, with candidates %llu, pairs listed %llu, blocks tested cell by cell %llu, pairs that met %llu; most pairs in a round %llu, most blocks one wave tested in a round %llu\n",
            c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
  }
  if (getenv("WFM_DEBUG") && hint_retries) fprintf(stderr, "[wfm] score hints: %llu roots ran past their hint and were run again without it\n", (unsigned long long)hint_retries);
  (void)hinted_roots;
  if (getenv("WFM_DEBUG") && band_jobs) fprintf(stderr, "[wfm] narrow rings: %llu jobs, %llu ran out of their band and were run again on full rings\n", (unsigned long long)band_jobs, (unsigned long long)band_retries);
  const auto t_levels = std::chrono::steady_clock::now();

  // ---- gather RLE pieces ----
  std::vector<int64_t> poff(n), pcap(n);
  for (size_t i = 0; i < n; ++i) { poff[i] = S->meta[first + i].rle_off; pcap[i] = (int64_t)S->meta[first + i].plen + S->meta[first + i].tlen; }
  if (h->i64a.ensure(n) || h->i64b.ensure(n) || h->i64c.ensure(n) || h->i32a.ensure(n) || h->total.ensure(1)) {
    h->err = "out of device memory"; return WFM_E_NOMEM;
  }
  HIPCHK(h, hipMemcpyAsync(h->i64a.p, poff.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->i64b.p, pcap.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemsetAsync(h->total.p, 0, sizeof(unsigned long long), h->stream));
  launch_compact(h->rle.p, h->i64a.p, h->i64b.p, h->rle_out.p, h->total.p, h->i64c.p, h->i32a.p, (int)n, h->stream);
  HIPCHK(h, hipGetLastError());
  std::vector<int64_t> ostart(n);
  std::vector<int32_t> ocount(n);
  unsigned long long total = 0;
  HIPCHK(h, hipMemcpyAsync(ostart.data(), h->i64c.p, n * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(ocount.data(), h->i32a.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&total, h->total.p, sizeof(total), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<uint32_t> runs((size_t)total + 1);
  if (total) HIPCHK(h, hipMemcpy(runs.data(), h->rle_out.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost));

  // ---- expand to op strings ----
  static const char opc[4] = {'M', 'X', 'I', 'D'};
  size_t arena_pos = runs_out ? runs_out->size() : arena_base;
  if (runs_out) runs_out->reserve(runs_out->size() + (size_t)total);
  int failed = 0;
  uint64_t cells_total = 0;
  for (size_t i = 0; i < n; ++i) {
    const size_t gi = first + i;  // problem id
    wfm_result_t& r = out[gi];
    r.status = prob_status[gi];
    r.cells = prob_cells[gi];
    cells_total += prob_cells[gi];
    r.ops_off = arena_pos; r.ops_len = 0; r.n_runs = 0; r.score = -1;
    if (r.status != WFM_ST_OK) { ++failed; continue; }
    const uint32_t* e = runs.data() + ostart[i];
    const int cnt = ocount[i];
    int64_t score = 0;
    uint64_t pc = 0, tc = 0;
    uint32_t nruns = 0;
    size_t pos = arena_pos;
    int k = 0;
    while (k < cnt) {
      const int op = (int)(e[k] & 3u);
      uint64_t len = e[k] >> 2;
      int k2 = k + 1;
      while (k2 < cnt && (int)(e[k2] & 3u) == op) { len += e[k2] >> 2; ++k2; }
      if (runs_out) {
        if (len >= (1u << 30)) { h->err = "run too long"; return WFM_E_ARG; }
        runs_out->push_back((uint32_t)(len << 2) | (uint32_t)op);
      } else {
        if (pos + len > arena_bytes) { h->err = "ops arena too small"; return WFM_E_ARENA; }
        memset(ops_arena + pos, opc[op], (size_t)len);
        pos += (size_t)len;
      }
      ++nruns;
      if (op == OP_X) { score += (int64_t)len * pen->x; pc += len; tc += len; }
      else if (op == OP_M) { pc += len; tc += len; }
      else {
        score += std::min<int64_t>(pen->o1 + (int64_t)len * pen->e1, pen->o2 + (int64_t)len * pen->e2);
        if (op == OP_I) tc += len; else pc += len;
      }
      k = k2;
    }
    if (pc != (uint64_t)S->meta[gi].plen || tc != (uint64_t)S->meta[gi].tlen) {
      if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] problem %zu: CIGAR spans %llu x %llu, sequences %d x %d\n", gi, (unsigned long long)pc, (unsigned long long)tc, S->meta[gi].plen, S->meta[gi].tlen);
      r.status = WFM_ST_UNREACHABLE;  // internal inconsistency: never report a broken CIGAR as ok
      ++failed;
      if (runs_out) runs_out->resize(arena_pos);
      continue;
    }
    if (runs_out) {
      // ops spelled: every M / X op advances both sequences, I the text, D the pattern
      uint64_t both = 0;
      for (size_t q = arena_pos; q < runs_out->size(); ++q) if (((*runs_out)[q] & 3u) <= (uint32_t)OP_X) both += (*runs_out)[q] >> 2;
      r.ops_len = (uint32_t)(pc + tc - both);
      pos = runs_out->size();
    } else {
      r.ops_len = (uint32_t)(pos - arena_pos);
    }
    r.n_runs = nruns;
    r.score = (int32_t)score;
    arena_pos = pos;
  }
  cells_total += h->stats.cells_tile;
  h->stats.cells = cells_total;
  uint64_t range_bases = 0;
  for (size_t i = first; i < last; ++i) range_bases += (uint64_t)S->meta[i].plen + (uint64_t)S->meta[i].tlen;
  h->stats.bytes_algorithmic = 48ull * cells_total + range_bases;
  h->stats.ms_breakpoint = tm.bp_ms;
  h->stats.ms_tile = tm.tile_ms;
  h->stats.ms_base = tm.base_ms;
  h->stats.ms_kernels = tm.bp_ms + tm.base_ms;
  h->stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  if (getenv("WFM_DEBUG"))
    fprintf(stderr, "[wfm] wall: total %.2f ms | levels %.2f (tile phase %.2f incl. kernels %.2f; base phase %.2f incl. kernels %.2f; bp kernels %.2f) | gather+expand %.2f\n",
            h->stats.ms_total, std::chrono::duration<double, std::milli>(t_levels - t_start).count(), wall_tile, tm.tile_ms, wall_base, tm.base_ms,
            tm.bp_ms - tm.tile_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_levels).count());
  return failed;
}

}  // namespace

extern "C" {

int wfm_device_count(void) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return ndev < 0 ? 0 : ndev;
}

int wfm_create(int device, wfm_handle_t** out) {
  if (!out) return WFM_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WFM_E_NODEVICE;
  if (device < 0 || device >= ndev) return WFM_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return WFM_E_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return WFM_E_HIP;
  // The kernels are written for the GFX9 / CDNA wave model and nothing else: DPP wave_shr / wave_shl / row_bcast, 64-lane waves, and waves that have
  // ended dropping out of s_barrier (wfa_tile2.hip lets the waves a narrow tile does not need return before its first barrier).  Any other
  // target fails here, loudly, instead of hanging in a kernel; wfm_selftest_dpp checks both properties on the device itself.
  if (strncmp(prop.gcnArchName, "gfx9", 4) != 0 || prop.warpSize != 64) return WFM_E_NODEVICE;
  wfm_handle* h = new wfm_handle();
  h->device = device;
  // (hipDeviceProp_t::name is empty on some boxes of the pool: the architecture name alone then)
  h->name = prop.name[0] ? std::string(prop.name) + " (" + prop.gcnArchName + ")" : std::string(prop.gcnArchName);
  if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return WFM_E_HIP; }
  (void)hipEventCreate(&h->ev0); (void)hipEventCreate(&h->ev1); (void)hipEventCreate(&h->ev2); (void)hipEventCreate(&h->ev3);
  h->tile_ev.resize(64);
  for (auto& e : h->tile_ev) (void)hipEventCreate(&e);
  (void)hipEventCreate(&h->ev_base);
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { fr = (size_t)16 << 30; }
  // the device's heap (dev_cache.hip) exists from the first handle on, with its first gigabytes mapped: no call of either path meets memory
  // the process has never had before it has used those (the budgets below are taken from what was free BEFORE, the heap's pool is theirs)
  wfm_dcache_warm();
  // 40 % of the free HBM, but no more than 32 GB: on this driver a first hipMalloc beyond a few tens of GB costs 35-40 ms
  // per GB (64 GB: 2.5-4.4 s, 110 GB: 3.9 s, 16 GB: 0.3 ms -- scripts/micro/malloc_cost.hip, profiles/r3_cold_start.md), which a
  // one-shot run pays in full: LPA all-vs-all (C2) aligned in 3.3 s cold and 0.33 s warm with rings sized for 115 GB.  A
  // level that needs more is worked off in chunks and on narrow rings
  // Handles of one device share it (the align driver keeps up to three per device): a further handle takes its 40 % of what
  // is free divided by the handles that are there already, so that on a smaller GPU the budgets together stay inside the memory
  int live = 0;
  { std::lock_guard<std::mutex> lk(g_base_mu); if (device < 64) live = g_dev_handles[device]++; }
  h->mem_budget = std::min<size_t>((size_t)((double)fr * 0.40 / (double)(1 + live)), (size_t)32 << 30);
  const char* env = getenv("WFM_MEM_BUDGET_MB");
  if (env) h->mem_budget = (size_t)atoll(env) << 20;
  h->mem_budget_full = h->mem_budget;
  *out = h;
  return WFM_OK;
}

void wfm_destroy(wfm_handle_t* h) {
  if (!h) return;
  for (wfm_handle* p : h->peers) wfm_destroy(p);
  h->peers.clear();
  bool last_of_process = false;
  if (!h->is_peer) {
    std::lock_guard<std::mutex> lk(g_base_mu);
    if (h->device < 64 && g_dev_handles[h->device] > 0) --g_dev_handles[h->device];
    last_of_process = true;
    for (int d = 0; d < 64; ++d) last_of_process &= g_dev_handles[d] == 0;
  }
  (void)hipSetDevice(h->device);
  h->ring.release(); h->base32.release(); h->base8.release(); h->rle.release(); h->rle_out.release();
  h->tilejobs.release(); h->tiletasks.release(); h->tilemak.release();
  h->revjobs.release(); h->bndjobs.release(); h->bndres.release();
  if (h->stage) { (void)hipHostFree(h->stage); h->stage = nullptr; h->stage_cap = 0; }
  h->p2rows.release(); h->p2max.release(); h->p2bmax.release(); h->p2pbmax.release(); h->p2jobs.release();
  h->bpjobs.release(); h->bpres.release(); h->bsjobs.release(); h->bsres.release();
  h->b2tjobs.release(); h->b2ttasks.release(); h->b2tkeys.release(); h->b2toffs.release(); h->b2tactive.release();
  h->i64a.release(); h->i64b.release(); h->i64c.release(); h->i32a.release(); h->seqflags.release(); h->flagjobs.release(); h->total.release();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->ev3) (void)hipEventDestroy(h->ev3);
  for (auto& e : h->tile_ev) if (e) (void)hipEventDestroy(e);
  if (h->ev_base) (void)hipEventDestroy(h->ev_base);
  if (h->attachment && h->attachment_free) h->attachment_free(h->attachment);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  // the last handle of the process: the sequence stores a map call left open for the call after it (host/fasta.cpp, keep_until_next: up to
  // 32 GB of host memory) are let go.  The device block cache stays -- a process that creates and destroys handles in turn (the test-suite)
  // would pay every block's first hipMalloc again -- and goes back to the driver with wfm_trim_device_cache(), see INTEGRATION.md.
  if (last_of_process && g_last_handle_hook) {
    // (counted again under the lock, and the hook runs under it: a wfm_create on another thread in between keeps the stores its map call is about to ask for)
    std::lock_guard<std::mutex> lk(g_base_mu);
    bool still_last = true;
    for (int d = 0; d < 64; ++d) still_last &= g_dev_handles[d] == 0;
    if (still_last) g_last_handle_hook();
  }
}

const char* wfm_last_error(const wfm_handle_t* h) { return h ? h->err.c_str() : "null handle"; }

void wfm_set_concurrent_calls(wfm_handle_t* h, int other_calls) { if (h) h->other_calls = other_calls > 0 ? other_calls : 0; }

size_t wfm_get_problem_flags(const wfm_handle_t* h, uint32_t* out, size_t n) {
  if (!h) return 0;
  const size_t have = h->prob_flags.empty() ? 0 : h->prob_flags.size() - 1;
  if (out) for (size_t i = 0; i < n && i < have; ++i) out[i] = h->prob_flags[i];
  return have;
}

int wfm_device_name(const wfm_handle_t* h, char* buf, size_t buflen) {
  if (!h || !buf || !buflen) return WFM_E_ARG;
  snprintf(buf, buflen, "%s", h->name.c_str());
  return WFM_OK;
}

size_t wfm_align_arena_bytes(const wfm_problem_t* problems, size_t n) {
  size_t t = 0;
  for (size_t i = 0; i < n; ++i) t += (size_t)problems[i].plen + (size_t)problems[i].tlen + 1;
  return t;
}

int wfm_upload_sequences(wfm_handle_t* h, const wfm_problem_t* problems, size_t n, wfm_seqset_t** out) {
  if (!h || !out || (n && !problems)) return WFM_E_ARG;
  *out = nullptr;
  HIPCHK(h, hipSetDevice(h->device));
  wfm_seqset* S = new wfm_seqset();
  S->meta.resize(n);
  size_t bytes = SEQ_PAD, rev_bytes = 0;
  int64_t rle = 0;
  for (size_t i = 0; i < n; ++i) {
    const wfm_problem_t& p = problems[i];
    if (p.plen < 0 || p.tlen < 0 || (int64_t)p.plen + p.tlen > (1 << 29) || (p.plen && !p.pattern) || (p.tlen && !p.text)) {
      delete S; h->err = "bad problem"; return WFM_E_ARG;
    }
    ProbMeta& m = S->meta[i];
    m.plen = p.plen; m.tlen = p.tlen; m.mode = p.mode;
    m.hint = (p.mode == WFM_MODE_END2END_BIWFA && p.score_hint > 0) ? p.score_hint : 0;
    m.pbf = std::min(std::max(p.pattern_begin_free, 0), p.plen); m.pef = std::min(std::max(p.pattern_end_free, 0), p.plen);
    m.tbf = std::min(std::max(p.text_begin_free, 0), p.tlen);    m.tef = std::min(std::max(p.text_end_free, 0), p.tlen);
    if (p.mode != WFM_MODE_ENDSFREE) { m.pbf = m.pef = m.tbf = m.tef = 0; }
    // forward copies first (the only part that crosses PCIe), the reversed copies of the BiWFA problems behind them
    m.p_fwd = (int64_t)bytes; bytes += (size_t)p.plen + SEQ_PAD;
    m.t_fwd = (int64_t)bytes; bytes += (size_t)p.tlen + SEQ_PAD;
    const bool need_rev = (p.mode == WFM_MODE_END2END_BIWFA);
    if (need_rev) {
      m.p_rev = (int64_t)rev_bytes; rev_bytes += (size_t)p.plen + SEQ_PAD;   // relative to the end of the forward part for now
      m.t_rev = (int64_t)rev_bytes; rev_bytes += (size_t)p.tlen + SEQ_PAD;
    } else { m.p_rev = -1; m.t_rev = -1; }
    m.rle_off = rle; rle += (int64_t)p.plen + p.tlen + 1;
    S->seq_bases += (uint64_t)p.plen + (uint64_t)p.tlen;
  }
  bytes += SEQ_PAD;
  const size_t fwd_bytes = bytes;
  for (size_t i = 0; i < n; ++i) {
    ProbMeta& m = S->meta[i];
    if (m.p_rev < 0) { m.p_rev = m.p_fwd; m.t_rev = m.t_fwd; }
    else { m.p_rev += (int64_t)fwd_bytes; m.t_rev += (int64_t)fwd_bytes; }
  }
  bytes = fwd_bytes + rev_bytes + SEQ_PAD;
  // Only the forward sequences (with their zero padding) are written on the host and cross PCIe; the reversed copies
  // BiWFA's reverse direction reads are made on the device after the upload (half the bytes, no byte-wise host loop).
  // The forward part is assembled in a pinned staging buffer kept with the handle, by a few threads.
  if (h->stage_cap < fwd_bytes) {
    if (h->stage) (void)hipHostFree(h->stage);
    h->stage = nullptr; h->stage_cap = 0;
    const size_t want = fwd_bytes + fwd_bytes / 4 + (1 << 20);
    if (hipHostMalloc((void**)&h->stage, want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      h->stage = nullptr;
    } else h->stage_cap = want;
  }
  std::vector<uint8_t> pageable;
  uint8_t* host = h->stage;
  if (!host) { pageable.resize(fwd_bytes); host = pageable.data(); }
  {
    const int nt = (int)std::min<size_t>(8, std::max<size_t>(1, n / 64));
    auto fill = [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; ++i) {
        const wfm_problem_t& p = problems[i];
        const ProbMeta& m = S->meta[i];
        // sequence, then SEQ_PAD zero bytes (the extension reads up to 40 bytes past a sub-range end)
        if (p.plen) memcpy(host + m.p_fwd, p.pattern, (size_t)p.plen);
        memset(host + m.p_fwd + p.plen, 0, SEQ_PAD);
        if (p.tlen) memcpy(host + m.t_fwd, p.text, (size_t)p.tlen);
        memset(host + m.t_fwd + p.tlen, 0, SEQ_PAD);
      }
    };
    memset(host, 0, SEQ_PAD);
    memset(host + fwd_bytes - SEQ_PAD, 0, SEQ_PAD);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(fill, n * (size_t)t / nt, n * (size_t)(t + 1) / nt);
    fill(0, n / (size_t)nt);
    for (auto& t : th) t.join();
  }
  // (from the block cache, like the arenas: a batch's sequences are about as long as the last batch's, and the block it freed serves)
  if (wfm_dmalloc((void**)&S->d_seq, bytes) != hipSuccess) { (void)hipGetLastError(); delete S; h->err = "out of device memory (sequences)"; return WFM_E_NOMEM; }
  S->bytes = bytes;
  S->rle_total = rle;
  // the 2-bit mirror the tile kernel extends on (wfa_tile2.hip): a quarter of a byte per base, made on the device
  const int64_t pk_words = ((int64_t)bytes + 15) / 16;
  if (wfm_dmalloc((void**)&S->d_pk, (size_t)(pk_words + PK_PAD_WORDS) * 4) != hipSuccess) {
    (void)hipGetLastError();
    wfm_dfree(S->d_seq); delete S; h->err = "out of device memory (packed sequences)"; return WFM_E_NOMEM;
  }
  hipError_t e = hipMemcpyAsync(S->d_seq, host, fwd_bytes, hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(S->d_seq + bytes - SEQ_PAD, 0, SEQ_PAD, h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(S->d_pk + pk_words, 0, (size_t)PK_PAD_WORDS * 4, h->stream);
  if (e == hipSuccess) {
    std::vector<SeqRev> rv;
    rv.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      const ProbMeta& m = S->meta[i];
      if (problems[i].mode == WFM_MODE_END2END_BIWFA) rv.push_back(SeqRev{m.p_fwd, m.p_rev, m.t_fwd, m.t_rev, m.plen, m.tlen});
    }
    if (!rv.empty()) {
      if (h->revjobs.ensure(rv.size())) e = hipErrorOutOfMemory;
      if (e == hipSuccess) e = hipMemcpyAsync(h->revjobs.p, rv.data(), rv.size() * sizeof(SeqRev), hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) { launch_reverse(S->d_seq, h->revjobs.p, (int)rv.size(), SEQ_PAD, h->stream); e = hipGetLastError(); }
    }
    // the mirror of everything (forward and reversed copies), and which problems are pure ACGT
    std::vector<int32_t> flags(n, 1);
    std::vector<SeqRev> fj(n);
    for (size_t i = 0; i < n; ++i) { const ProbMeta& m = S->meta[i]; fj[i] = SeqRev{m.p_fwd, m.p_fwd, m.t_fwd, m.t_fwd, m.plen, m.tlen}; }
    if (e == hipSuccess && n && (h->seqflags.ensure(n) || h->flagjobs.ensure(n))) e = hipErrorOutOfMemory;
    if (e == hipSuccess && n) e = hipMemsetAsync(h->seqflags.p, 1, n * sizeof(int32_t), h->stream);
    if (e == hipSuccess && n) e = hipMemcpyAsync(h->flagjobs.p, fj.data(), n * sizeof(SeqRev), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
      launch_seq_pack(S->d_seq, S->d_pk, pk_words, (int64_t)bytes, h->flagjobs.p, (int)n, n ? h->seqflags.p : nullptr, h->stream);
      e = hipGetLastError();
    }
    if (e == hipSuccess && n) e = hipMemcpyAsync(flags.data(), h->seqflags.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // (fj is read by the copy above)
    if (e == hipSuccess) {
      S->acgt.assign(flags.begin(), flags.end());
      if (getenv("WFM_DEBUG") && atoi(getenv("WFM_DEBUG")) > 1) {
        size_t bad = 0;
        for (int32_t f : flags) bad += f == 0;
        fprintf(stderr, "[wfm] upload: %zu problems, %zu of them with something other than upper-case ACGT (byte kernels)\n", n, bad);
      }
    }
  }
  if (e != hipSuccess) { wfm_dfree(S->d_seq); wfm_dfree(S->d_pk); delete S; h->err = hipGetErrorString(e); return WFM_E_HIP; }
  *out = S;
  return WFM_OK;
}

void wfm_free_sequences(wfm_handle_t* h, wfm_seqset_t* s) {
  if (!s) return;
  if (h) (void)hipSetDevice(h->device);
  // back to the block cache; one wait for the device serves both (hipFree waited once per block)
  if (s->d_seq || s->d_pk) (void)hipDeviceSynchronize();
  if (s->d_seq) wfm_dfree_nosync(s->d_seq);
  if (s->d_pk) wfm_dfree_nosync(s->d_pk);
  delete s;
}

namespace {
// both output forms: ops_arena (one byte per op) or, with runs != nullptr, one malloc'd buffer of merged runs
int align_resident_any(wfm_handle_t* h, const wfm_penalties_t* pen, wfm_seqset_t* s, wfm_result_t* out,
                       char* ops_arena, size_t arena_bytes, uint32_t** runs, size_t* n_runs_total) {
  if (!h || !s || !out || (!runs && !ops_arena && arena_bytes)) return WFM_E_ARG;
  if (runs) *runs = nullptr;
  if (n_runs_total) *n_runs_total = 0;
  const size_t n = s->meta.size();
  std::vector<std::vector<uint32_t>> part_runs(1);
  // the parts' runs side by side in one buffer; ops_off of part k's problems shifted by what lies before it
  auto hand_over_runs = [&](const std::vector<size_t>& cut) -> int {
    if (!runs) return WFM_OK;
    size_t total = 0;
    for (const auto& v : part_runs) total += v.size();
    uint32_t* buf = (uint32_t*)malloc((total + 1) * sizeof(uint32_t));
    if (!buf) { h->err = "out of host memory (CIGAR runs)"; return WFM_E_NOMEM; }
    size_t at = 0;
    for (size_t k = 0; k < part_runs.size(); ++k) {
      if (!part_runs[k].empty()) memcpy(buf + at, part_runs[k].data(), part_runs[k].size() * sizeof(uint32_t));
      if (at) for (size_t i = cut[k]; i < cut[k + 1]; ++i) out[i].ops_off += at;
      at += part_runs[k].size();
    }
    *runs = buf;
    if (n_runs_total) *n_runs_total = total;
    return WFM_OK;
  };
  const bool overlap = !(getenv("WFM_OVERLAP") && atoi(getenv("WFM_OVERLAP")) == 0);  // read per call: bench.py times both forms
  if (hipSetDevice(h->device) != hipSuccess || hipEventRecord(h->ev_base, h->stream) != hipSuccess ||
      hipEventSynchronize(h->ev_base) != hipSuccess) { h->err = "hipEventRecord failed"; return WFM_E_HIP; }
  h->call_base = h->ev_base;
  h->prob_flags.assign(n + 1, 0u);  // (the parts of a call each write their own problems' entries)
  h->tile_iv.clear(); h->bp_iv.clear(); h->base_iv.clear();
  auto busy_ms = [](std::vector<std::pair<float, float>> iv) {  // length of the union of the intervals
    std::sort(iv.begin(), iv.end());
    double total = 0, lo = 0, hi = -1;
    for (const auto& x : iv) {
      if (x.first > hi) { if (hi > lo) total += hi - lo; lo = x.first; hi = x.second; }
      else hi = std::max<double>(hi, x.second);
    }
    if (hi > lo) total += hi - lo;
    return total;
  };
  h->mem_budget = h->mem_budget_full;
  auto keep_abs = [&](std::vector<std::pair<float, float>> iv) {  // the union as intervals on the device's own clock
    h->busy_abs.clear();
    double moved = 0;
    hipEvent_t db = device_base_event(h->device, &moved);
    float off_f = 0;
    const hipError_t ee = db ? hipEventElapsedTime(&off_f, db, h->ev_base) : hipErrorInvalidValue;
    const double off = moved + (double)off_f;
    if (ee != hipSuccess) {
      (void)hipGetLastError();
      if (getenv("WFM_DEBUG")) fprintf(stderr, "[wfm] busy intervals: no common clock (%s)\n", hipGetErrorString(ee));
      return;
    }
    std::sort(iv.begin(), iv.end());
    double lo = 0, hi = -1;
    for (const auto& x : iv) {
      if (x.first > hi) { if (hi > lo) h->busy_abs.emplace_back(off + lo, off + hi); lo = x.first; hi = x.second; }
      else hi = std::max<double>(hi, x.second);
    }
    if (hi > lo) h->busy_abs.emplace_back(off + lo, off + hi);
  };
  auto finish_single = [&](int rc) {
    h->stats.ms_tile_busy = busy_ms(h->tile_iv);
    h->stats.ms_bp_busy = busy_ms(h->bp_iv);
    h->stats.ms_base_busy = busy_ms(h->base_iv);
    std::vector<std::pair<float, float>> all = h->tile_iv;
    all.insert(all.end(), h->bp_iv.begin(), h->bp_iv.end());
    all.insert(all.end(), h->base_iv.begin(), h->base_iv.end());
    h->stats.ms_any_busy = busy_ms(all);
    keep_abs(all);
    h->stats.streams = 1;
    return rc;
  };
  auto run_single = [&]() {
    const int rc = finish_single(align_resident_impl(h, pen, s, 0, n, out, ops_arena, arena_bytes, 0, runs ? &part_runs[0] : nullptr, h->prob_flags.data()));
    if (rc < 0) return rc;
    const int hrc = hand_over_runs(std::vector<size_t>{0, n});
    return hrc != WFM_OK ? hrc : rc;
  };
  if (!overlap || n < 8) return run_single();
  // Parts of the batch side by side, each with its own stream and arenas (peer handles on the same device)
  // and its own host thread: while one part sits in the few-workgroup levels of the step kernel or waits for
  // the host, the other parts' tiles fill the machine.  Problems are independent, the parts only share the
  // (read-only) sequences and the caller's output buffers.
  static const int want = [] { const char* e = getenv("WFM_STREAMS"); return e ? std::max(1, std::min(8, atoi(e))) : 3; }();  // measured: 2 -> 124, 3 -> 120, 4 -> 160 ms on C3
  // every part needs room for its own arenas: no split below 256 MB per part
  size_t parts = std::min<size_t>(std::min<size_t>((size_t)want, n / 4), h->mem_budget_full >> 28);
  // A batch of hundreds of problems fills the device on its own -- its levels are thousands of workgroups wide -- and the
  // align driver keeps further batches in flight on handles of their own: such a batch runs as one part
  // -- when its problems come with score hints, i.e. from a driver that knows them to be near-identical records.  Hundreds of
  // problems nobody has said anything about (the strong-scaling bench at N = 1: 512 pairs at 5 %) are deep, run in many
  // chunks of full rings, and gain from parts as 64 of them do (512 pairs: 1136 ms as one part)
  if (!getenv("WFM_STREAMS") && n >= 512) {
    size_t hinted = 0, biwfa = 0;  // (patch calls -- ends-free problems only -- stay one part)
    for (size_t i = 0; i < n; ++i) { biwfa += s->meta[i].mode == WFM_MODE_END2END_BIWFA; hinted += s->meta[i].hint > 0; }
    // ... and only then: a batch that has the device to itself (a mapping file of one batch: LPA all-vs-all, a scaled pangenome rank) is a
    // chain of short launches per level and chunk, and three such chains side by side took its device time from 80 to 58 ms (C2) and from
    // 66 to 49 ms (scaled C4 rank) -- gpurun_out/r5b_ab.log, r5f_ab.log.  Patch calls are chains as well (budget 256 -> 1020 -> the rest).
    if (h->other_calls > 0 && (biwfa == 0 || hinted * 2 >= biwfa)) parts = 1;
  }
  if (!getenv("WFM_STREAMS") && parts > 2) {
    // a batch whose full rings would not fit the budget -- thousands of long records, which then run on narrow
    // rings -- is bound by the host's work between the many small launches: measured best with two parts
    // (C4-like records, 60 Mbp of queries: 2 -> 2.6 s, 3 -> 3.5 s)
    size_t ring_bytes = 0;
    for (size_t i = 0; i < n && ring_bytes <= h->mem_budget_full; ++i)
      ring_bytes += ((size_t)s->meta[i].plen + (size_t)s->meta[i].tlen + 9) * 2 * 5 * RING * 2 * 4;
    if (ring_bytes > h->mem_budget_full) parts = 2;
  }
  if (parts < 2) return run_single();
  while (h->peers.size() + 1 < parts) {
    wfm_handle_t* p = nullptr;
    if (wfm_create(h->device, &p) != WFM_OK) break;
    p->is_peer = true;
    { std::lock_guard<std::mutex> lk(g_base_mu); if (h->device < 64 && g_dev_handles[h->device] > 0) --g_dev_handles[h->device]; }
    h->peers.push_back(p);
  }
  const size_t np = h->peers.size() + 1;
  h->mem_budget = h->mem_budget_full / np;  // the parts share the primary handle's budget
  for (wfm_handle* pk : h->peers) pk->mem_budget = h->mem_budget;
  // contiguous parts of equal WFA cost, sum of (plen + tlen)^2
  std::vector<double> cost(n + 1, 0.0);
  for (size_t i = 0; i < n; ++i) {
    const double l = (double)s->meta[i].plen + (double)s->meta[i].tlen;
    cost[i + 1] = cost[i] + l * l;
  }
  std::vector<size_t> cut(np + 1, n);
  cut[0] = 0;
  for (size_t k = 1, i = 0; k < np; ++k) {
    while (i < n && cost[i] < cost[n] * (double)k / (double)np) ++i;
    cut[k] = std::min(std::max(i, cut[k - 1] + 1), n - (np - k));
  }
  std::vector<size_t> base(np, 0);
  for (size_t k = 1; k < np; ++k) {
    base[k] = base[k - 1];
    for (size_t i = cut[k - 1]; i < cut[k]; ++i) base[k] += (size_t)s->meta[i].plen + (size_t)s->meta[i].tlen + 1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<int> rcs(np, 0);
  std::vector<std::thread> th;
  part_runs.resize(np);
  for (size_t k = 1; k < np; ++k) {
    wfm_handle* pk = h->peers[k - 1];
    pk->call_base = h->ev_base;
    pk->tile_iv.clear(); pk->bp_iv.clear(); pk->base_iv.clear();
    th.emplace_back([&, k, pk] { rcs[k] = align_resident_impl(pk, pen, s, cut[k], cut[k + 1], out, ops_arena, arena_bytes, base[k], runs ? &part_runs[k] : nullptr, h->prob_flags.data()); });
  }
  rcs[0] = align_resident_impl(h, pen, s, cut[0], cut[1], out, ops_arena, arena_bytes, 0, runs ? &part_runs[0] : nullptr, h->prob_flags.data());
  for (auto& t : th) t.join();
  int failed = 0;
  for (size_t k = 0; k < np; ++k) {
    if (rcs[k] < 0) { if (k) h->err = h->peers[k - 1]->err; return rcs[k]; }
    failed += rcs[k];
  }
  {
    const int hrc = hand_over_runs(cut);
    if (hrc != WFM_OK) return hrc;
  }
  wfm_stats_t& a = h->stats;
  std::vector<std::pair<float, float>> iv = h->tile_iv, ivb = h->bp_iv, ivs = h->base_iv;
  for (size_t k = 1; k < np; ++k) {
    const wfm_stats_t& b = h->peers[k - 1]->stats;
    a.cells += b.cells; a.bytes_algorithmic += b.bytes_algorithmic; a.ms_kernels += b.ms_kernels; a.ms_breakpoint += b.ms_breakpoint;
    a.ms_base += b.ms_base; a.levels = std::max(a.levels, b.levels); a.bp_jobs += b.bp_jobs; a.base_jobs += b.base_jobs;
    a.bp_launches += b.bp_launches; a.base_launches += b.base_launches; a.cells_bp += b.cells_bp; a.cells_base += b.cells_base;
    a.p2_launches += b.p2_launches; a.p2_jobs += b.p2_jobs; a.p2_more += b.p2_more;
    a.cells_tile += b.cells_tile; a.ms_tile += b.ms_tile; a.tile_launches += b.tile_launches; a.tile_tasks += b.tile_tasks;
    a.cells_tile_unique += b.cells_tile_unique;
    iv.insert(iv.end(), h->peers[k - 1]->tile_iv.begin(), h->peers[k - 1]->tile_iv.end());
    ivb.insert(ivb.end(), h->peers[k - 1]->bp_iv.begin(), h->peers[k - 1]->bp_iv.end());
    ivs.insert(ivs.end(), h->peers[k - 1]->base_iv.begin(), h->peers[k - 1]->base_iv.end());
  }
  a.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  a.ms_tile_busy = busy_ms(iv);
  a.ms_bp_busy = busy_ms(ivb);
  a.ms_base_busy = busy_ms(ivs);
  iv.insert(iv.end(), ivb.begin(), ivb.end());
  iv.insert(iv.end(), ivs.begin(), ivs.end());
  a.ms_any_busy = busy_ms(iv);
  keep_abs(iv);
  a.streams = (uint32_t)np;
  return failed;
}

int align_batch_any(wfm_handle_t* h, const wfm_penalties_t* pen, const wfm_problem_t* problems, size_t n,
                    wfm_result_t* out, char* ops_arena, size_t arena_bytes, uint32_t** runs, size_t* n_runs_total);
}  // namespace

int wfm_align_resident(wfm_handle_t* h, const wfm_penalties_t* pen, wfm_seqset_t* s, wfm_result_t* out,
                       char* ops_arena, size_t arena_bytes) {
  return align_resident_any(h, pen, s, out, ops_arena, arena_bytes, nullptr, nullptr);
}

int wfm_align_resident_rle(wfm_handle_t* h, const wfm_penalties_t* pen, wfm_seqset_t* s, wfm_result_t* out,
                           uint32_t** runs, size_t* n_runs_total) {
  if (!runs) return WFM_E_ARG;
  return align_resident_any(h, pen, s, out, nullptr, 0, runs, n_runs_total);
}

int wfm_align_batch(wfm_handle_t* h, const wfm_penalties_t* pen, const wfm_problem_t* problems, size_t n,
                    wfm_result_t* out, char* ops_arena, size_t arena_bytes) {
  return align_batch_any(h, pen, problems, n, out, ops_arena, arena_bytes, nullptr, nullptr);
}

int wfm_align_batch_rle(wfm_handle_t* h, const wfm_penalties_t* pen, const wfm_problem_t* problems, size_t n,
                        wfm_result_t* out, uint32_t** runs, size_t* n_runs_total) {
  if (!runs) return WFM_E_ARG;
  return align_batch_any(h, pen, problems, n, out, nullptr, 0, runs, n_runs_total);
}

void wfm_free_runs(uint32_t* runs) { free(runs); }

// Self-test of the arenas' growth policy (DevBuf::ensure): capacities after ensure(n0), ensure(cap + 1), ensure(what fits): out3[0..2] in elements.
// A regrowth must at least double (every regrowth is a fresh hipMalloc at 30 - 70 ms per GB), a request that fits must not allocate.
int wfm_selftest_arena_growth(wfm_handle_t* h, size_t n0, size_t* out3) {
  if (!h || !out3 || n0 == 0) return WFM_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  DevBuf<int32_t> b;
  if (b.ensure(n0)) return WFM_E_NOMEM;
  out3[0] = b.cap;
  if (b.ensure(b.cap + 1)) { b.release(); return WFM_E_NOMEM; }
  out3[1] = b.cap;
  const int32_t* before = b.p;
  if (b.ensure(b.cap - 1) || b.p != before) { b.release(); return WFM_E_HIP; }
  out3[2] = b.cap;
  b.release();
  return WFM_OK;
}

int wfm_selftest_dpp(wfm_handle_t* h, int32_t* out128) {
  if (!h || !out128) return WFM_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  return selftest_dpp(out128, h->stream) == 0 ? WFM_OK : WFM_E_HIP;
}

int wfm_score_bounds(wfm_handle_t* h, const wfm_penalties_t* pen, const wfm_problem_t* problems, size_t n, int32_t* out) {
  if (!h || !pen || !out || (n && !problems)) return WFM_E_ARG;
  if (n == 0) return WFM_OK;
  wfm_seqset_t* S = nullptr;
  int rc = wfm_upload_sequences(h, problems, n, &S);
  if (rc != WFM_OK) return rc;
  std::vector<BoundJob> bj(n);
  for (size_t i = 0; i < n; ++i) bj[i] = BoundJob{S->meta[i].p_fwd, S->meta[i].t_fwd, S->meta[i].plen, S->meta[i].tlen};
  auto run = [&]() -> int {
    if (h->bndjobs.ensure(n) || h->bndres.ensure(n)) { h->err = "out of device memory (score bounds)"; return WFM_E_NOMEM; }
    HIPCHK(h, hipMemcpyAsync(h->bndjobs.p, bj.data(), n * sizeof(BoundJob), hipMemcpyHostToDevice, h->stream));
    launch_bound(S->d_seq, h->bndjobs.p, h->bndres.p, (int)n, DevPen{pen->x, pen->o1, pen->e1, pen->o2, pen->e2}, h->stream);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->bndres.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return WFM_OK;
  };
  rc = run();
  wfm_free_sequences(h, S);
  return rc;
}

namespace {
int align_batch_any(wfm_handle_t* h, const wfm_penalties_t* pen, const wfm_problem_t* problems, size_t n,
                    wfm_result_t* out, char* ops_arena, size_t arena_bytes, uint32_t** runs, size_t* n_runs_total) {
  if (!h) return WFM_E_ARG;
  if (runs) *runs = nullptr;
  if (n_runs_total) *n_runs_total = 0;
  wfm_seqset_t* S = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = wfm_upload_sequences(h, problems, n, &S);
  if (rc != WFM_OK) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  rc = align_resident_any(h, pen, S, out, ops_arena, arena_bytes, runs, n_runs_total);
  const auto t2 = std::chrono::steady_clock::now();
  wfm_free_sequences(h, S);
  if (getenv("WFM_DEBUG"))
    fprintf(stderr, "[wfm] align_batch: %zu problems, upload %.2f ms, align %.2f ms, free %.2f ms\n", n,
            std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
  return rc;
}
}  // namespace

size_t wfm_get_busy_intervals(const wfm_handle_t* h, double* start_end_ms, size_t cap) {
  if (!h) return 0;
  const size_t n = h->busy_abs.size();
  for (size_t i = 0; i < n && i < cap && start_end_ms; ++i) { start_end_ms[2 * i] = h->busy_abs[i].first; start_end_ms[2 * i + 1] = h->busy_abs[i].second; }
  return n;
}

int wfm_get_stats(const wfm_handle_t* h, wfm_stats_t* out) {
  if (!h || !out) return WFM_E_ARG;
  *out = h->stats;
  return WFM_OK;
}

}  // extern "C"

#include "wfa_handle.h"
hipStream_t wfm_stream(wfm_handle_t* h) { return h->stream; }
int wfm_device(const wfm_handle_t* h) { return h->device; }
void wfm_set_error(wfm_handle_t* h, const std::string& msg) {
  static std::mutex mu;  // several host threads may work on one handle (the device winnower next to the hashing thread)
  std::lock_guard<std::mutex> lk(mu);
  h->err = msg;
}
void* wfm_attachment(wfm_handle_t* h) { return h->attachment; }
void wfm_set_attachment(wfm_handle_t* h, void* p, void (*destroy)(void*)) {
  if (h->attachment && h->attachment_free && h->attachment != p) h->attachment_free(h->attachment);
  h->attachment = p;
  h->attachment_free = destroy;
}
