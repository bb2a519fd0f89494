// wfa_tile2.hip -- the register time tile of wfa_kernels.hip (wfa_tile_reg_kernel) with the two things its SQ counters asked
// for (profiles/r3_sq.json: 197 VALU + 127 SALU instructions per wave and score step for 2 x 64 cells, VALU busy 0.55):
//
//  * the extension reads 2-BIT PACKED sequences.  wfm_upload_sequences keeps a packed mirror of the sequence buffer (base at
//    byte index a = bits 2 (a & 15) .. of word a >> 4, code (c >> 1) & 3: A 0, C 1, T 2, G 3); a problem whose sequences hold
//    anything but upper-case ACGT (an N, soft-masked bases) is flagged and stays on the byte kernel, whose comparisons are exact
//    for any alphabet.  One probe is 16 bases = two words per sequence, one v_alignbit each, one xor, one ffbl -- where the byte
//    form needed three words and two alignbits per sequence for 8 bases; the 8 KB windows in LDS now hold 32 k bases of each
//    sequence (a 100-score block of a 0.1 % record crosses 20 kb), and the wave-cooperative tail of a long run compares
//    64 lanes x 32 bases = 2048 bases per round trip instead of 512.
//  * no range bookkeeping in the step.  A cell outside the triangle |k| <= s is NULL by induction (its sources are), a column
//    outside [-pl, tl] or cut off by the score bound is nulled when it is WRITTEN (one compare against a per-cell constant:
//    the last score at which the cell is inside), and a cell inside only ever reads cells that were inside at their own score
//    (wfa_kernels.hip, Rng) -- so nothing has to be selected when a value is READ: the five closed-form source ranges per step
//    (50 SALU) and the 18 selects per thread of the non-interior threads are gone.  The snapshot load still filters by range
//    (the ring holds stale cells outside), and so do all readers of what the tile writes.
//
// Same contract as wfa_tile_reg_kernel (snapshot in -> T steps -> snapshot out + per-step maxima; P2: every row kept), same
// results bit for bit (tests/test_align_gpu.py runs every case on both kernels).  With 64 threads the workgroup is one wave:
// neighbours by DPP wave shifts only, no mailbox, and the barriers compile to nothing -- the form for the narrow wavefronts
// of near-identical records (C2 / C4), where a step is a chain of latencies and not a matter of throughput.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "wfa_device.h"
#include "wfa_pack.h"

// WFM_TILE_TRACE (a build switch of its own, never the shipped library: scripts/tile_trace.sh): s_memtime stamps of one step's phases, per wave, of the
// first 48 workgroups of every launch of more than 1024 tiles; the last such launch stays in g_tile_trace.
#ifdef WFM_TILE_TRACE
__device__ unsigned long long g_tile_trace[48 * 16 * 128 * 4];
extern "C" int wfm_debug_tile_trace(unsigned long long* out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_trace), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#if WFM_TILE_TRACE == 1
#define WFM_TRACE_STAMP(slot_) do { if (trace_on) { const unsigned long long ts_ = __builtin_readcyclecounter(); if (lane == 0) g_tile_trace[(((size_t)blockIdx.x * 16 + wv) * 128 + (t & 127)) * 4 + (slot_)] = ts_; } } while (0)
#else
#define WFM_TRACE_STAMP(slot_) do { } while (0)
#endif
// WFM_TILE_TRACE=2: one stamp per ten-step body and one each at the kernel's begin, before the step loop, after it and at the end (slots 120 .. 123 of
// the wave's row): the split of a tile's life into snapshot load, steps and snapshot store without the per-step stamps' own cost
#define WFM_TRACE_MARK(idx_) do { if (WFM_TILE_TRACE == 2 && trace_on2) { const unsigned long long ts_ = __builtin_readcyclecounter(); if (lane == 0) g_tile_trace[(((size_t)blockIdx.x * 16 + wv) * 128 + (idx_)) * 4] = ts_; } } while (0)
#else
#define WFM_TRACE_STAMP(slot_) do { } while (0)
#define WFM_TRACE_MARK(idx_) do { } while (0)
#endif

namespace wfm {

namespace {

constexpr int PK_WIN_DW = 2048;               // words per sequence window in LDS
constexpr int PK_WIN_BASES = PK_WIN_DW * 16;  // 32768 bases
constexpr int PK_SLACK_DW = 8;                // words past the window a probe may touch
constexpr int TAIL_DIRECT_MAX = 2;            // pk_extend2<true>: up to this many lanes with a run past 16 bases go to the wave's tail at once

// A workgroup barrier that orders LDS only: __syncthreads() also waits for every global store of the wave to be acknowledged (vmcnt(0)), which a step
// loop that writes rows nobody reads before the kernel's end pays for at every step
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// two adjacent 32-bit values at a dword-aligned address as one access (global loads and stores of 8 bytes need no more than that)
struct __attribute__((packed, aligned(4))) Pair32 { int x, y; };
__device__ __forceinline__ void ld_pair(const int32_t* p, int& a, int& b) { const Pair32 v = *reinterpret_cast<const Pair32*>(p); a = v.x; b = v.y; }
__device__ __forceinline__ void st_pair(int32_t* p, int a, int b) { Pair32 v; v.x = a; v.y = b; *reinterpret_cast<Pair32*>(p) = v; }
struct Rng2 { int pl, tl, kb_lo, kb_hi; };
__device__ __forceinline__ Rng2 make_rng2(int pl, int tl, int sub) { Rng2 r; r.pl = pl; r.tl = tl; r.kb_lo = (tl - pl) - sub; r.kb_hi = (tl - pl) + sub; return r; }
__device__ __forceinline__ int rng2_lo(const Rng2& r, int s) { return max(max(-r.pl, -s), r.kb_lo + s); }
__device__ __forceinline__ int rng2_hi(const Rng2& r, int s) { return min(min(r.tl, s), r.kb_hi - s); }
constexpr int RNG2_BACK = 25;  // = RNG_BACK of wfa_kernels.hip
__device__ __forceinline__ void rng2_block(const Rng2& r, int s_from, int s_to, int& L, int& R) {
  L = max(max(-r.pl, -s_to), r.kb_lo + s_from - RNG2_BACK);
  R = min(min(r.tl, s_to), r.kb_hi - s_from + RNG2_BACK);
}

__device__ __forceinline__ int rdl(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// lane i <- lane i - 1 (lane 0 keeps NULL) / lane i <- lane i + 1 (lane 63 keeps NULL): one VALU instruction each
__device__ __forceinline__ int from_prev_lane(int x) { return __builtin_amdgcn_update_dpp(WF_NULL, x, 0x138, 0xf, 0xf, false); }  // wave_shr:1
__device__ __forceinline__ int from_next_lane(int x) { return __builtin_amdgcn_update_dpp(WF_NULL, x, 0x130, 0xf, 0xf, false); }  // wave_shl:1
// the same with whatever for the edge lane (zero): for tiles of several waves, whose edge lanes take their values from the mailbox anyway -- no
// register has to be set to NULL before every shift
__device__ __forceinline__ int from_prev_lane0(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int from_next_lane0(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true); }

__device__ __forceinline__ int wave_max63(int x) {
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));  // row_shr:1
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));  // row_shr:2
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));  // row_shr:4
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));  // row_shr:8
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));  // row_bcast:15
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return x;
}

// explicit address spaces: the LDS windows and the global mirror are read by the same code shapes, and a pointer that may be
// either would be read with flat loads
typedef const __attribute__((address_space(3))) uint32_t* lds_words;
typedef const __attribute__((address_space(1))) uint32_t* glb_words;

// Where a tile's sequences are read: the LDS windows while the offsets (counted from the windows' origin) stay inside, the
// global mirror from the same origin beyond
struct PkSrc {
  lds_words lP, lT;
  glb_words gP, gT;
};

// bases 16 .. 79 after a probe that matched 16: four more words per sequence.  Returns the run length so far (16 .. 80).
__device__ __forceinline__ int pk_stage2(const PkSrc& S, unsigned oP, unsigned oT) {
  uint32_t wa[5], wb[5];
  if (max(oP, oT) <= (unsigned)(PK_WIN_BASES - 96)) {
    const lds_words a = S.lP + (oP >> 4) + 1;
    const lds_words b = S.lT + (oT >> 4) + 1;
#pragma unroll
    for (int q = 0; q < 5; ++q) { wa[q] = a[q]; wb[q] = b[q]; }
  } else {
    const glb_words a = S.gP + (oP >> 4) + 1;
    const glb_words b = S.gT + (oT >> 4) + 1;
#pragma unroll
    for (int q = 0; q < 5; ++q) { wa[q] = a[q]; wb[q] = b[q]; }
  }
  const unsigned sa = oP << 1, sb = oT << 1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t x = alignbit32(wa[q + 1], wa[q], sa) ^ alignbit32(wb[q + 1], wb[q], sb);
    if (x) return 16 + 16 * q + (int)(__builtin_ctz(x) >> 1);
  }
  return 80;
}

// Runs that go on past 80 bases are finished by the whole wave, one pending lane after the other: 64 lanes x 32 bases per
// round trip.  Every lane of the wave makes the call (pend = false: nothing of its own).
__device__ __forceinline__ int pk_wave_tail(const PkSrc& S, unsigned oP, unsigned oT, int n, int maxn, bool pend) {
  unsigned long long todo = __ballot(pend);
  const int lane = (int)(threadIdx.x & 63u);
  while (todo) {
    const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
    todo &= todo - 1;
    const unsigned p0 = (unsigned)rdl((int)oP, src), t0 = (unsigned)rdl((int)oT, src);
    const int mx = rdl(maxn, src);
    int nn = rdl(n, src);
    int res;
    for (;;) {
      const int off = nn + lane * 32;
      const bool past = off >= mx;
      uint64_t x = 0;
      if (!past) {
        const unsigned a = p0 + (unsigned)off, b = t0 + (unsigned)off;
        if (max(a, b) <= (unsigned)(PK_WIN_BASES - 48)) x = pk32(S.lP, a) ^ pk32(S.lT, b);
        else x = pk32(S.gP, a) ^ pk32(S.gT, b);
      }
      const unsigned long long hit = __ballot(past || x != 0);
      if (hit) {
        const int f = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit));
        const unsigned xlo = (unsigned)rdl((int)(uint32_t)x, f), xhi = (unsigned)rdl((int)(uint32_t)(x >> 32), f);
        const uint64_t xf = ((uint64_t)xhi << 32) | xlo;
        const int at = nn + f * 32;
        res = at >= mx ? mx : min(mx, at + (xf ? (int)(__builtin_ctzll(xf) >> 1) : 0));
        break;
      }
      nn += 2048;
    }
    if (lane == src) n = res;
  }
  return n;
}

// The extension of the two cells of a lane: m[c] >= 0 is the cell's offset (else: no cell), oP / oT the offsets of its
// (v, h) from the windows' origin, maxn how far the sequences go; ext[c] = the number of bases that agree (<= maxn).
// 16 bases of every cell at once from the LDS windows (an offset that has left them is clamped into them for the probe and
// done again from the global mirror; cells that hold nothing probe whatever their garbage offset clamps to), then
// 64 more for the cells whose 16 agreed, then the wave-cooperative tail.  Every lane of the wave makes the call.
// (MASKED: the probe's window words are addressed by the offset's bits 4 .. 14 instead of by the clamped offset -- one instruction less per
// sequence; an offset beyond the window probes a word it does not mean, and is probed again from the global mirror as before.  maxn of a cell
// that holds nothing may be anything: every use of it is behind `live`.)
__device__ __forceinline__ uint32_t pk16_win(lds_words w, unsigned o) {
  typedef const __attribute__((address_space(3))) char* lds_bytes;
  const lds_words q = (lds_words)((lds_bytes)w + ((o >> 2) & (unsigned)((PK_WIN_DW - 1) * 4)));
  // (the bit offset as o + o: v_lshlrev_b32 issues at 4 cycles per wave on gfx950, v_add_u32 at 2 -- profiles/r6_valu_issue.md; the compiler
  // turns any source form of the doubling back into the shift)
  unsigned sh;
  asm("v_add_u32 %0, %1, %1" : "=v"(sh) : "v"(o));
  return alignbit32(q[1], q[0], sh);
}
template <bool MASKED = false>
__device__ __forceinline__ void pk_extend2(const PkSrc& SRC, const int (&m)[2], const unsigned (&oP)[2], const unsigned (&oT)[2], const int (&maxn)[2], int (&ext)[2],
                                           bool tail_direct = false) {
  bool more[2], outw = false;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool live = m[c] >= 0;
    const unsigned qa = min(oP[c], (unsigned)(PK_WIN_BASES - 1)), qb = min(oT[c], (unsigned)(PK_WIN_BASES - 1));
    const uint32_t x = MASKED ? (pk16_win(SRC.lP, oP[c]) ^ pk16_win(SRC.lT, oT[c])) : (pk16(SRC.lP, qa) ^ pk16(SRC.lT, qb));
    const unsigned n16 = first_diff16(x);
    ext[c] = min((int)n16, maxn[c]);
    more[c] = live && n16 >= 16u && maxn[c] > 16;
    outw |= live && max(oP[c], oT[c]) > (unsigned)(PK_WIN_BASES - 1);
  }
  if (__any(outw)) {  // rare: the probe again from the global mirror for the cells beyond the windows
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (m[c] >= 0 && max(oP[c], oT[c]) > (unsigned)(PK_WIN_BASES - 1)) {
        const uint32_t x = pk16(SRC.gP, oP[c]) ^ pk16(SRC.gT, oT[c]);
        const unsigned n16 = first_diff16(x);
        ext[c] = min((int)n16, maxn[c]);
        more[c] = n16 >= 16u && maxn[c] > 16;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const unsigned long long mm = __ballot(more[c]);
    if (mm) {
      bool tail = false;
      // One or two lanes whose 16 bases agreed, in a job of near-identical sequences (tail_direct: the job's score bound is under a sixteenth
      // of its length, wave-uniform), where such a run is hundreds of bases long: the wave takes it from base 16 on at once instead of giving the
      // lane its 64 bases first -- one LDS round trip less on the step's chain (scaled C4 rank 46.3 -> 44.8 ms).  Divergent sequences (44 % of
      // the cells at 5 % have 16 bases that agree, and their runs end within the next 64): the lanes look at those side by side as before --
      // without the job test C3 lost 2.6 % to the sparsely filled waves at the tiles' edges.
      if (MASKED && tail_direct && __popcll(mm) <= (unsigned)TAIL_DIRECT_MAX) tail = more[c];
      else if (more[c]) {
        const int n = pk_stage2(SRC, oP[c], oT[c]);
        ext[c] = min(n, maxn[c]);
        tail = n >= 80 && maxn[c] > 80;
      }
      // runs longer than 80 bases: the wave finishes them together (uniform control flow: every lane is here)
      if (__any(tail)) ext[c] = pk_wave_tail(SRC, oP[c], oT[c], ext[c], maxn[c], tail);
    }
  }
}


// Round 6: the same staged extension for the FAST form with the flag bookkeeping taken out of the step's vector stream (profiles/r6_valu_issue.md:
// compares and selects issue at 4 cycles per wave like everything but add / sub / logic, and the round-5 form spent 14 of its 103 vector
// instructions per step on flags: live, 16 bases agree, more than 16 left, beyond the window -- each its own compare per cell -- and on turning the
// two `more` masks into 0 / 1 registers and back because they were alive across the rare branch).  Here: a cell goes on past its probe when it is
// live and its clipped extension is 16 (one compare; a cell with exactly 16 bases left makes a harmless visit to stage 2, which clips again);
// beyond the window <=> m > lim (one signed compare against a per-cell constant, false for a cell that holds nothing); the two masks stay
// wave-level bit masks and a lane looks its own bit up inside the rare branches only (__builtin_amdgcn_ballot_w64 on the compare itself: HIP's
// __ballot goes through a 0 / 1 register and a second compare).
// (round 6, the pipelined step: the probe's straight-line part -- addresses, window words, first difference, the wave's two masks -- apart from
// the rare branches that follow it, so that the step's recurrences can stand between the two in ONE basic block and overlap the probe's LDS
// round trip: pk_probe2_fast, then pk_rare2_fast.)
struct PkProbe { unsigned long long mm[2]; bool outw; };
__device__ __forceinline__ PkProbe pk_probe2_fast(const PkSrc& SRC, const int (&m)[2], const unsigned (&oP)[2], const unsigned (&oT)[2], const int (&maxn)[2],
                                                  const int (&lim)[2], int (&ext)[2]) {
  PkProbe R;
  R.outw = false;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t x = pk16_win(SRC.lP, oP[c]) ^ pk16_win(SRC.lT, oT[c]);
    ext[c] = min((int)first_diff16(x), maxn[c]);
    R.mm[c] = __builtin_amdgcn_ballot_w64(m[c] >= 0) & __builtin_amdgcn_ballot_w64(ext[c] >= 16);  // (one ballot of the conjunction goes through a 0 / 1 register)
    R.outw |= m[c] > lim[c];
  }
  return R;
}
__device__ __forceinline__ void pk_rare2_fast(const PkSrc& SRC, PkProbe R, const int (&m)[2], const unsigned (&oP)[2], const unsigned (&oT)[2], const int (&maxn)[2],
                                              const int (&lim)[2], int (&ext)[2], bool tail_direct) {
  if (__builtin_expect(__any(R.outw), 0)) {  // rare: the probe again from the global mirror for the cells beyond the windows
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bool more = m[c] >= 0 && ext[c] >= 16;
      if (m[c] > lim[c]) {
        const uint32_t x = pk16(SRC.gP, oP[c]) ^ pk16(SRC.gT, oT[c]);
        ext[c] = min((int)first_diff16(x), maxn[c]);
        more = ext[c] >= 16;
      }
      R.mm[c] = __builtin_amdgcn_ballot_w64(more);
    }
  }
  const unsigned lane = threadIdx.x & 63u;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (__builtin_expect(R.mm[c] != 0, 0)) {
      const bool more = (R.mm[c] >> lane) & 1ull;
      bool tail = false;
      if (tail_direct && __popcll(R.mm[c]) <= (unsigned)TAIL_DIRECT_MAX) tail = more;  // (pk_extend2: a lone long run of a near-identical job goes to the wave at once)
      else if (more) {
        const int n = pk_stage2(SRC, oP[c], oT[c]);
        ext[c] = min(n, maxn[c]);
        tail = n >= 80 && maxn[c] > 80;
      }
      if (__any(tail)) ext[c] = pk_wave_tail(SRC, oP[c], oT[c], ext[c], maxn[c], tail);
    }
  }
}
__device__ __forceinline__ void pk_extend2_fast(const PkSrc& SRC, const int (&m)[2], const unsigned (&oP)[2], const unsigned (&oT)[2], const int (&maxn)[2],
                                                const int (&lim)[2], int (&ext)[2], bool tail_direct) {
  const PkProbe R = pk_probe2_fast(SRC, m, oP, oT, maxn, lim, ext);
  pk_rare2_fast(SRC, R, m, oP, oT, maxn, lim, ext, tail_direct);
}

}  // namespace

// ---------------------------------------------------------------------------
// packed mirror of the sequence buffer + "is this problem pure ACGT" flags
// ---------------------------------------------------------------------------
// word i of the mirror = bytes 16 i .. 16 i + 15 of the buffer
__global__ __launch_bounds__(256) void seq_pack_kernel(const uint8_t* __restrict__ seq, uint32_t* __restrict__ pk, int64_t nwords, int64_t nbytes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  uint32_t v = 0;
  if ((i + 1) * 16 <= nbytes) {
    const uint4 q = *reinterpret_cast<const uint4*>(seq + i * 16);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t c = (w[j] >> 1) & 0x03030303u;  // the code of each of the four bytes
      v |= ((c & 3u) | ((c >> 6) & 0xcu) | ((c >> 12) & 0x30u) | ((c >> 18) & 0xc0u)) << (8 * j);
    }
  } else {
    for (int j = 0; j < 16; ++j) {
      const int64_t a = i * 16 + j;
      const uint32_t c = a < nbytes ? (uint32_t)seq[a] : 0u;
      v |= pack_code((uint8_t)c) << (2 * j);
    }
  }
  pk[i] = v;
}
// one workgroup per (problem, pattern / text): flag[problem] = 0 when a byte is not one of A C G T
__global__ __launch_bounds__(256) void seq_acgt_kernel(const uint8_t* __restrict__ seq, const SeqRev* __restrict__ jobs, int32_t* __restrict__ flag) {
  const SeqRev J = jobs[blockIdx.x >> 1];
  const bool text = blockIdx.x & 1;
  const uint8_t* src = seq + (text ? J.t_fwd : J.p_fwd);
  const int n = text ? J.tlen : J.plen;
  bool bad = false;
  for (int q = threadIdx.x; q < n; q += blockDim.x) {
    const uint8_t c = src[q];
    bad |= !pack_is_acgt(c);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) flag[blockIdx.x >> 1] = 0;
}
void launch_seq_pack(const uint8_t* seq, uint32_t* pk, int64_t nwords, int64_t nbytes, const SeqRev* jobs, int njobs, int32_t* flag, hipStream_t st) {
  if (nwords > 0) hipLaunchKernelGGL(seq_pack_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, seq, pk, nwords, nbytes);
  if (njobs > 0) hipLaunchKernelGGL(seq_acgt_kernel, dim3((unsigned)njobs * 2), dim3(256), 0, st, seq, jobs, flag);
}

// ---------------------------------------------------------------------------
// the tile kernel
// ---------------------------------------------------------------------------
// FAST (round 5, the default; WFM_TILE_FAST=0 runs the round-4 form for A/B): what profiles/r4_sq.json's 152 vector + 56 scalar instructions per
// wave and step could lose without touching a result --
//  * the five "is this offset inside the problem" selects of a cell are only made by a wave in which some cell's largest source has left the
//    problem (h > tl or v > pl: the far edges of the matrix, which the two directions of a BiWFA job never reach together); otherwise every
//    source is either inside or NULL already.  A NULL then drifts -- NULL + 1 + ... -- instead of being set back to WF_NULL at every step:
//    every reader of a wavefront takes "negative" for NULL (max(), >= 0, o0 + o1 >= tl), and -2^30 + 17 per step stays negative for 6 * 10^7 steps;
//  * the per-step maximum of the antidiagonals goes to a slot of the wave's own (plain store) instead of through an LDS atomic (a scalar loop);
//  * the mailbox's buffer index is the step's parity within the ten-step body, at compile time (adjacent steps never share a buffer);
//  * the probe's window words are addressed by masking (pk_extend2<true>), dead cells keep what the arithmetic leaves them with.
// FINE (round 6, FAST phase-1 form only): the per-score maxima of the antidiagonals -- a six-step DPP reduction per wave and score, a seventh of
// the step's vector issue -- are kept only by the FINE instantiation, which takes the tiles of the jobs whose block needs them (TileJob::fine_s:
// the block in which the directions meet); the other instantiation takes all other tiles and keeps one running maximum per lane, reduced once
// per block.  Both are launched over the same task list (the host leaves out the one no job of the block can need); a tile that is not this
// instantiation's ends before its first load.  (One kernel with a runtime test was built first: the optimizer then neither unrolls the ten-step
// body nor keeps the delay lines in registers; two copies of the loop in one kernel spill.)
template <int NTMAX, bool P2, bool FAST, bool FINE = true>
__global__ __launch_bounds__(NTMAX) void wfa_tile2_kernel(const uint32_t* __restrict__ pk, int32_t* __restrict__ ring_arena,
                                                        const TileJob* __restrict__ jobs, const TileTask* __restrict__ tasks,
                                                        int32_t* __restrict__ mak_out, int T, int32_t* __restrict__ p2_arena, int coarse) {
  constexpr int C = 2, LB = 25, H = LB + 1, NCL = 5, DEP = 6, E1 = 2;
  constexpr bool WAVE1 = NTMAX == 64;  // one wave: no mailbox; __syncthreads() is a wave barrier for a 64-thread workgroup
  // PIPE (round 6): the extension of a step's row is made one step later.  Nothing reads M[s] before step s + 5 (its own diagonal's mismatch
  // source; the neighbours read it at s + 10 and s + 25), while the step's chain -- barrier, mailbox, recurrences, probe, first difference, the
  // ballots' branches, publish, barrier -- was 1300 cycles long whether one workgroup or two sat on the CU (scripts/tile_trace2.py).  Now the probe
  // of row s - 1 is issued right behind the mailbox read of step s and the recurrences of step s run during its LDS round trip: two independent
  // chains in one basic block instead of one chain of twice the length.  The row waits in pM[] (before its extension); its delay line takes it
  // one step late, which no reader can see (a class's line is only read at its own steps).
  constexpr bool PIPE = FAST && !P2;
  // mailbox of the wave edges: [parity][slot][side][value].  Side 0 of slot w holds what lane 63 of wave w - 1 hands to lane 0 of wave w, side 1 of
  // slot w what lane 0 of wave w hands to lane 63 of wave w - 1; slot 0's side 0 and the slot behind the last wave are never written and stay
  // NULL, so the edge lanes of a tile read their mailbox like all others and the wave shifts need no NULL to fall back on
  __shared__ __attribute__((aligned(16))) int s_edge[2][WAVE1 ? 1 : 17][2][4];
  __shared__ __attribute__((aligned(16))) uint32_t s_winP[PK_WIN_DW + PK_SLACK_DW], s_winT[PK_WIN_DW + PK_SLACK_DW];
  __shared__ int s_wlo[2];
  extern __shared__ __attribute__((aligned(16))) int s_makr[];  // [T + 1]; FAST: [waves][T + 1], a row per wave
  TileTask tk = tasks[blockIdx.x];
  const TileJob J = jobs[tk.job];
  if (!J.active) return;
  // score of the snapshot this direction starts from (mode 6: the block before the job's s0 once more, from the snapshot before it -- TileJob::ring_prev)
  const int sbase = P2 ? (tk.dir == 0 ? J.tf : J.tr) : (J.mode == 6 ? J.s0 - T : J.s0);
  const Rng2 RG = make_rng2(J.pl, J.tl, J.sub);
  const bool lowdiv = (J.packed & 2) != 0;  // near-identical sequences: the job's known score is under a sixteenth of its length (the host says: pk_extend2, tail_direct)
  int halo = T;  // columns computed on either side of the core (the trapezoid loses one per step)
  {
    const int s1 = sbase + T;
    int L, R;
    rng2_block(RG, sbase, s1, L, R);
    const int idx = tk.core_lo, core = tk.core_hi;
    tk.core_lo = L + idx * core;
    tk.core_hi = min(R, tk.core_lo + core - 1);
    if (tk.core_lo > R) return;
    if (tk.core_lo == L && tk.core_hi == R) halo = 0;  // one tile for the whole range: nothing beside it to take from
  }
  const int dir = tk.dir, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kA = tk.core_lo - halo;
  // A launch is sized for its widest tile; the waves a narrower one does not need end here, before the first barrier (waves that
  // have ended drop out of s_barrier: scripts/micro/barrier_exit.hip), and give their SIMD slots to other tiles.  In a pangenome
  // level half the lanes of a launch held no cell (20 % at the deeper levels).
  const int nw = min((int)(blockDim.x >> 6), (tk.core_hi + halo - kA) / (64 * C) + 1), NT = nw * 64;
  if (wv >= nw) return;
#ifdef WFM_TILE_TRACE
  const bool trace_on2 = !P2 && FAST && !WAVE1 && gridDim.x > 1024 && blockIdx.x < 48 && sbase == 3000 && J.mode == 0;  // (one launch in the middle of a deep job's run)
#endif
  WFM_TRACE_MARK(120);
  if (tid < 2) s_wlo[tid] = INT32_MAX;
  if (!WAVE1) for (int i = tid; i < (int)(sizeof(s_edge) / sizeof(int)); i += NT) ((int*)s_edge)[i] = WF_NULL;
  const int64_t aP = dir == 0 ? J.p_fwd : J.p_rev, aT = dir == 0 ? J.t_fwd : J.t_rev;  // byte index of the sequences' first bases
  const int pl = J.pl, tl = J.tl, s0 = sbase;
  const int k0 = kA + tid * C;  // first diagonal of this thread
  const int64_t width = J.width;
  const int32_t* rin = ring_arena + ((!P2 && J.mode == 6) ? J.ring_prev : J.ring_in) + J.koff + (int64_t)dir * 5 * RING * width;
  int32_t* rout = ring_arena + J.ring_out + J.koff + (int64_t)dir * 5 * RING * width;
  const int kmax = tk.core_hi + halo;  // last diagonal of the tile
  const int Tn = (!P2 && J.mode == 1) ? (dir == 0 ? J.tf : J.tr) : T;
  // how many of the block's last rows of the gap components go into the output snapshot: all H where somebody may read that deep (the run up to
  // the meeting point, whose output phase 2 reads; the block that runs again for such a run's sake; jobs without a third ring), else the E1 rows
  // the next block loads
  const int t_stream = Tn - ((J.mode == 1 || J.mode == 6 || J.ring_prev < 0) ? H : E1);
  // per-score maxima of the antidiagonals only where the advance kernel reads them score by score (TileJob::fine_s): the block that runs again
  // because the directions met in it (mode 5), and the blocks from fine_s on -- the FINE instantiation's tiles.  Elsewhere one running maximum per
  // lane and ONE wave reduction per block (mode 1, the run up to the meeting point: none at all -- nobody reads its maxima)
  if (!P2 && FAST && !(coarse & 2)) {  // (bit 1: the only instantiation launched for this block -- the FINE one -- takes every tile)
    const bool fine = !coarse || J.mode == 5 || (J.mode == 0 && sbase + T >= J.fine_s);
    if (fine != FINE) return;
  }
  // (the running maximum of a lane lives in LDS, one ds_max_i32 per step into the lane's own word: as a register carried around the step loop it
  // kept the optimizer from unrolling the ten-step body -- the delay lines, indexed by the step's residue class, went to scratch memory)
  __shared__ int s_run[(!P2 && FAST && !FINE) ? NTMAX : 1];
  if (!P2 && FAST && !FINE) s_run[tid] = 0;

  // Mh[c][r][e] = M[sr - 5 e][k0+c], sr = the newest score <= current with (sr - s0) mod 5 == r
  int Mh[C][NCL][DEP];
  int I1h[C][E1], D1h[C][E1], I2h[C], D2h[C];
  // ---- snapshot load (rows <= s0): what lies outside a row's own range is NULL, whatever the ring holds there
  // (round 6: a lane's two diagonals are one 8-byte load -- dword alignment is all a global load needs; with one 4-byte load per diagonal every
  // load instruction used half of each cache line it touched and the lines came up from L2 twice, the tile's 26 rows of M for one diagonal lying
  // between the two uses)
  {
    const bool kin0 = k0 <= kmax, kin1 = k0 + 1 <= kmax;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < NCL; ++r)
#pragma unroll
        for (int e = 0; e < DEP; ++e) Mh[c][r][e] = WF_NULL;
    auto ld_row = [&](int comp, int sc, int& a, int& b) {
      const int lo = rng2_lo(RG, sc), hi = rng2_hi(RG, sc);
      const bool ok0 = kin0 && sc >= 0 && k0 >= lo && k0 <= hi, ok1 = kin1 && sc >= 0 && k0 + 1 >= lo && k0 + 1 <= hi;
      int x = WF_NULL, y = WF_NULL;
      if (ok0 || ok1) ld_pair(rin + ((int64_t)(comp * RING + (sc & RMASK))) * width + k0, x, y);
      a = ok0 ? x : WF_NULL; b = ok1 ? y : WF_NULL;
    };
#pragma unroll
    for (int d = 0; d < H; ++d)  // row s0-d: class (-d mod 5), the (d/5)-th newest of its class
      ld_row(C_M, s0 - d, Mh[0][(NCL - d % NCL) % NCL][d / NCL], Mh[1][(NCL - d % NCL) % NCL][d / NCL]);
#pragma unroll
    for (int d = 0; d < E1; ++d) {
      ld_row(C_I1, s0 - d, I1h[0][d], I1h[1][d]);
      ld_row(C_D1, s0 - d, D1h[0][d], D1h[1][d]);
    }
    ld_row(C_I2, s0, I2h[0], I2h[1]);
    ld_row(C_D2, s0, D2h[0], D2h[1]);
  }
  const int MKS = T + 1;  // stride of a wave's row of s_makr (FAST)
  if (FINE && !P2) for (int t = tid; t < (FAST && !WAVE1 ? nw * MKS : MKS); t += NT) s_makr[t] = 0;
  // ---- per-cell constants
  unsigned hmaxu[C];  // largest offset inside the problem on this diagonal: min(tl, pl + k)
  int s_last[C];      // the last score at which the cell is inside its row (columns outside [-pl, tl]: never)
  int negk[C];        // -k for cells of the core, else far below any 2 m
  bool incore[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = k0 + c;
    const bool colok = (k >= -pl) && (k <= tl);
    hmaxu[c] = colok ? (unsigned)min(tl, pl + k) : 0u;
    s_last[c] = colok ? min(k - RG.kb_lo, RG.kb_hi - k) : -1;
    incore[c] = k >= tk.core_lo && k <= tk.core_hi;
    negk[c] = incore[c] ? -k : -(1 << 29);
  }
  // ---- sequence windows: every offset this tile will ever extend from is >= the smallest live offset of its history
  {
    int hlo = INT32_MAX, vlo = INT32_MAX;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      int lo = INT32_MAX;
#pragma unroll
      for (int r = 0; r < NCL; ++r)
#pragma unroll
        for (int e = 0; e < DEP; ++e) lo = min(lo, Mh[c][r][e] >= 0 ? Mh[c][r][e] : INT32_MAX);
#pragma unroll
      for (int d = 0; d < E1; ++d) { lo = min(lo, I1h[c][d] >= 0 ? I1h[c][d] : INT32_MAX); lo = min(lo, D1h[c][d] >= 0 ? D1h[c][d] : INT32_MAX); }
      lo = min(lo, I2h[c] >= 0 ? I2h[c] : INT32_MAX);
      lo = min(lo, D2h[c] >= 0 ? D2h[c] : INT32_MAX);
      if (lo != INT32_MAX) { hlo = min(hlo, lo); vlo = min(vlo, lo - k); }
    }
    WFM_TRACE_MARK(124);  // (the loaded rows have been read: the snapshot is there)
    __syncthreads();  // s_wlo initialised
    if (hlo != INT32_MAX) { atomicMin(&s_wlo[0], hlo); atomicMin(&s_wlo[1], max(vlo, 0)); }
    __syncthreads();
  }
  WFM_TRACE_MARK(125);
  const int wT0 = s_wlo[0] == INT32_MAX ? 0 : s_wlo[0], wP0 = s_wlo[1] == INT32_MAX ? 0 : s_wlo[1];
  // window origins as absolute base indices, word aligned; offsets of a cell from them: oP = v + dP, oT = h + dT
  const int64_t oriP = (aP + wP0) & ~(int64_t)15, oriT = (aT + wT0) & ~(int64_t)15;
  const int dP = (int)(aP - oriP), dT = (int)(aT - oriT);
  PkSrc SRC;
  SRC.lP = (lds_words)s_winP; SRC.lT = (lds_words)s_winT;
  SRC.gP = (glb_words)pk + (oriP >> 4); SRC.gT = (glb_words)pk + (oriT >> 4);
  for (int i = tid; i < PK_WIN_DW + PK_SLACK_DW; i += NT) {  // (the mirror is padded by more than a window: no bound to check)
    s_winP[i] = SRC.gP[i];
    s_winT[i] = SRC.gT[i];
  }
  int cP[C];  // oP of cell c = m + cP[c]  (v = m - k)
  int wlim[C];  // FAST: the largest offset whose 16-base probe stays inside both windows (pk_extend2_fast)
#pragma unroll
  for (int c = 0; c < C; ++c) { cP[c] = dP - (k0 + c); wlim[c] = PK_WIN_BASES - 1 - max(cP[c], dT); }
  // (Tried in round 4, as in round 3 on the byte kernel, and taken out again: letting a wave skip the steps at which it holds no
  // cell -- before the triangle reaches its diagonals, 26 scores after the score bound has cut off its last one; two scalar
  // compares per step against wave-uniform bounds.  The kernel sits at its 128 registers: the extra control flow made it spill,
  // C3 83.8 -> 99.2 ms per step, C2's leg 84.6 -> 90.2 ms, the scaled C4 rank 72.1 -> 76.5.)
  __syncthreads();

  // FAST: ten steps per unrolled body.  A class's delay line is then visited twice per body: the first visit parks its new row in the
  // line's last entry (the row of score s - 30, which nothing reads any more) and shifts nothing, the second reads one entry further up
  // and shifts by two (six moves for two steps instead of ten); the two-deep I1 / D1 lines need no move at all (the row of score s - 2
  // sits where the new one goes: entry s & 1 of the body's own count).
  constexpr int UB = FAST ? 2 * NCL : NCL;
  // PIPE: the snapshot's newest row (score s0, class 0) becomes the pending row -- extending it again finds nothing to add -- and its line is put
  // into the state a class's line is in between its first visit of a body and the (late) completion of its second: the second visit of class 0
  // is what the body's first step completes
  int pM[C];
  if (PIPE) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      pM[c] = Mh[c][0][0];
      const int l1 = Mh[c][0][1], l2 = Mh[c][0][2], l3 = Mh[c][0][3], l4 = Mh[c][0][4], l5 = Mh[c][0][5];
      Mh[c][0][DEP - 1] = l1; Mh[c][0][0] = l2; Mh[c][0][1] = l3; Mh[c][0][2] = l4; Mh[c][0][3] = l5;
    }
  }
  // mailbox (FAST): where this lane's edge values go -- lane 63: side 0 of the slot of the wave above, lane 0: side 1 of its own wave's slot
  constexpr int EDGE_PLANE = (WAVE1 ? 1 : 17) * 2 * 4;
  const bool edge_lane = lane == 0 || lane == 63;
  int* const edge_wr = lane == 63 ? &s_edge[0][WAVE1 ? 0 : wv + 1][0][0] : &s_edge[0][WAVE1 ? 0 : wv][1][0];
  // the gap components of a step's row into the output snapshot (the block's last rows only).  (One 8-byte store per component for a lane's two
  // diagonals, as the M rows have it below, was built: the pair form and its fall-back for the lanes at a core's edge in every one of the ten
  // steps of the body cost 12 - 20 bytes of spills.)
  auto stream_gap_rows = [&](int s, const int (&vI1)[C], const int (&vI2)[C], const int (&vD1)[C], const int (&vD2)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (incore[c] && s <= s_last[c]) {
        const int64_t ro = ((int64_t)(s & RMASK)) * width + k0 + c;
        rout[(int64_t)C_I1 * RING * width + ro] = vI1[c];
        rout[(int64_t)C_I2 * RING * width + ro] = vI2[c];
        rout[(int64_t)C_D1 * RING * width + ro] = vD1[c];
        rout[(int64_t)C_D2 * RING * width + ro] = vD2[c];
      }
    }
  };
  // a finished row into its class's line (jjx: the step's number within the body, 1 .. UB)
  auto line_take = [&](auto JJX, const int (&row)[C]) {
    constexpr int jjx = decltype(JJX)::value;
    constexpr int clx = jjx % NCL;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (jjx <= NCL) Mh[c][clx][DEP - 1] = row[c];
      else {
        const int parked = Mh[c][clx][DEP - 1];
#pragma unroll
        for (int e = DEP - 1; e > 1; --e) Mh[c][clx][e] = Mh[c][clx][e - 2];
        Mh[c][clx][1] = parked;
        Mh[c][clx][0] = row[c];
      }
    }
  };
  WFM_TRACE_MARK(121);
  // (Tried in round 6 and taken out: a second copy of the body without the tests for the block's end and for the streaming of the gap rows,
  // for the bodies that lie whole before the block's last H scores -- five scalar instructions less in 74 of 100 steps, and 164 bytes of spills.)
  for (int tb = 0; tb < Tn; tb += UB) {
    WFM_TRACE_MARK(tb / UB);
#pragma unroll
  for (int jj = 1; jj <= UB; ++jj) {
    const int t = tb + jj;
    if (t > Tn) break;
    const int cl = jj % NCL;  // residue class of this step's score: compile time after unrolling
    const bool vB = jj > NCL;  // (FAST) the class's second visit in this body
    const int e1x = FAST ? (jj & 1) : E1 - 1;  // where the I1 / D1 row of score s - 2 sits
    const int s = s0 + t;
    // rows this step reads from its class: s-5, s-10, s-25 = entries 0, 1, 4 of the line (second visit: the parked row, entries 0 and 3)
#define M_S5(c_) (vB ? Mh[c_][cl][DEP - 1] : Mh[c_][cl][0])
#define M_S10(c_) (vB ? Mh[c_][cl][0] : Mh[c_][cl][1])
#define M_S25(c_) (vB ? Mh[c_][cl][3] : Mh[c_][cl][4])
    int lM10, lM25, lI1, lI2, rM10, rM25, rD1, rD2;
#ifdef WFM_TILE_TRACE
    const bool trace_on = !P2 && FAST && !WAVE1 && gridDim.x > 1024 && blockIdx.x < 48;
#endif
    WFM_TRACE_STAMP(3);  // the step begins (the previous one's bookkeeping is issued)
    if (!WAVE1) {
      const int par = FAST ? (jj & 1) : (t & 1);  // (a body of ten steps: its own count alternates across bodies as well)
      // publish the wave-edge history values needed by the neighbouring waves in this step
      if (FAST) {
        // (round 6) one store for both edges -- lane 63 hands its left-bound values to the slot of the wave above, lane 0 its right-bound ones to
        // its own slot: four selects, one exec mask, one ds_write_b128 -- and the reads without a mask: every lane reads both neighbour slots (uniform
        // addresses: broadcasts), and the wave shifts take what was read as the value the edge lane KEEPS (update_dpp's `old`: lane 0 has no
        // source under wave_shr, lane 63 none under wave_shl).  17 instructions where the masked form had 24.
        typedef int __attribute__((ext_vector_type(4))) i4;
        if (edge_lane) {
          i4 w;
          w.x = lane == 63 ? M_S10(C - 1) : M_S10(0);
          w.y = lane == 63 ? M_S25(C - 1) : M_S25(0);
          w.z = lane == 63 ? I1h[C - 1][e1x] : D1h[0][e1x];
          w.w = lane == 63 ? I2h[C - 1] : D2h[0];
          *reinterpret_cast<i4*>(edge_wr + par * EDGE_PLANE) = w;
        }
        __syncthreads();
        WFM_TRACE_STAMP(0);  // the barrier has let go
        const i4 La = *reinterpret_cast<const i4*>(&s_edge[par][wv][0][0]);
        const i4 Ra = *reinterpret_cast<const i4*>(&s_edge[par][wv + 1][1][0]);
        lM10 = __builtin_amdgcn_update_dpp(La.x, M_S10(C - 1), 0x138, 0xf, 0xf, false);
        lM25 = __builtin_amdgcn_update_dpp(La.y, M_S25(C - 1), 0x138, 0xf, 0xf, false);
        lI1 = __builtin_amdgcn_update_dpp(La.z, I1h[C - 1][e1x], 0x138, 0xf, 0xf, false);
        lI2 = __builtin_amdgcn_update_dpp(La.w, I2h[C - 1], 0x138, 0xf, 0xf, false);
        rM10 = __builtin_amdgcn_update_dpp(Ra.x, M_S10(0), 0x130, 0xf, 0xf, false);
        rM25 = __builtin_amdgcn_update_dpp(Ra.y, M_S25(0), 0x130, 0xf, 0xf, false);
        rD1 = __builtin_amdgcn_update_dpp(Ra.z, D1h[0][e1x], 0x130, 0xf, 0xf, false);
        rD2 = __builtin_amdgcn_update_dpp(Ra.w, D2h[0], 0x130, 0xf, 0xf, false);
      } else {
      if (lane == 63) { int* e = s_edge[par][wv + 1][0]; e[0] = M_S10(C - 1); e[1] = M_S25(C - 1); e[2] = I1h[C - 1][e1x]; e[3] = I2h[C - 1]; }
      if (lane == 0)  { int* e = s_edge[par][wv][1];     e[0] = M_S10(0);     e[1] = M_S25(0);     e[2] = D1h[0][e1x];     e[3] = D2h[0]; }
      __syncthreads();
      WFM_TRACE_STAMP(0);  // the barrier has let go
      lM10 = from_prev_lane0(M_S10(C - 1)); lM25 = from_prev_lane0(M_S25(C - 1));
      lI1 = from_prev_lane0(I1h[C - 1][e1x]); lI2 = from_prev_lane0(I2h[C - 1]);
      rM10 = from_next_lane0(M_S10(0)); rM25 = from_next_lane0(M_S25(0));
      rD1 = from_next_lane0(D1h[0][e1x]); rD2 = from_next_lane0(D2h[0]);
      if (lane == 0)  { const int* e = s_edge[par][wv][0];     lM10 = e[0]; lM25 = e[1]; lI1 = e[2]; lI2 = e[3]; }
      if (lane == 63) { const int* e = s_edge[par][wv + 1][1]; rM10 = e[0]; rM25 = e[1]; rD1 = e[2]; rD2 = e[3]; }
      }
    } else {
      lM10 = from_prev_lane(M_S10(C - 1)); lM25 = from_prev_lane(M_S25(C - 1));
      lI1 = from_prev_lane(I1h[C - 1][e1x]); lI2 = from_prev_lane(I2h[C - 1]);
      rM10 = from_next_lane(M_S10(0)); rM25 = from_next_lane(M_S25(0));
      rD1 = from_next_lane(D1h[0][e1x]); rD2 = from_next_lane(D2h[0]);
    }
    int nM[C], nI1[C], nI2[C], nD1[C], nD2[C], nMis[C];
    // PIPE: the probe of the row the step before left pending, ahead of this step's recurrences
    int pext[C], pmaxn[C];
    unsigned poP[C], poT[C];
    PkProbe PR;
    if (PIPE) {
#pragma unroll
      for (int c = 0; c < C; ++c) { poP[c] = (unsigned)(pM[c] + cP[c]); poT[c] = (unsigned)(pM[c] + dT); pmaxn[c] = (int)hmaxu[c] - pM[c]; }
      PR = pk_probe2_fast(SRC, pM, poP, poT, pmaxn, wlim, pext);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int a10 = c == 0 ? lM10 : M_S10(c - 1), b10 = c == C - 1 ? rM10 : M_S10(c + 1);
      const int a25 = c == 0 ? lM25 : M_S25(c - 1), b25 = c == C - 1 ? rM25 : M_S25(c + 1);
      const int i1 = c == 0 ? lI1 : I1h[c - 1][e1x], d1 = c == C - 1 ? rD1 : D1h[c + 1][e1x];
      const int i2 = c == 0 ? lI2 : I2h[c - 1], d2 = c == C - 1 ? rD2 : D2h[c + 1];
      const int mx = M_S5(c);
      // in-bounds <=> 0 <= offset <= min(tl, pl + k)   (h <= tl and h - k <= pl)
      const unsigned hm = hmaxu[c];
      int ins1 = max(a10, i1) + 1, ins2 = max(a25, i2) + 1, del1 = max(b10, d1), del2 = max(b25, d2), mis = mx + 1;
      if (!FAST) {
        ins1 = (unsigned)ins1 <= hm ? ins1 : WF_NULL;
        ins2 = (unsigned)ins2 <= hm ? ins2 : WF_NULL;
        del1 = (unsigned)del1 <= hm ? del1 : WF_NULL;
        del2 = (unsigned)del2 <= hm ? del2 : WF_NULL;
        mis = (unsigned)mis <= hm ? mis : WF_NULL;
      }
      nI1[c] = ins1; nI2[c] = ins2; nD1[c] = del1; nD2[c] = del2; nMis[c] = mis;
      nM[c] = max(max(max(ins1, ins2), mis), max(del1, del2));
    }
    if (FAST) {
      // some cell of the wave whose largest source lies beyond the problem, or whose column is outside [-pl, tl] or cut off by the score
      // bound at this score: the selects of the round-4 form, for the whole wave (a wave inside the problem and inside the bound: none)
      // (Tried in round 6 and taken out: "some cell is past its last score" against the wave's smallest s_last, a scalar compare instead of two
      // vector compares -- the second branch cost more than the compares: C3 62.5 -> 63.2 ms per batch.)
      bool special = false;
#pragma unroll
      for (int c = 0; c < C; ++c) special |= (nM[c] > (int)hmaxu[c]) | (s > s_last[c]);
      if (__builtin_expect(__any(special), 0)) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const unsigned hm = hmaxu[c];
          nI1[c] = (unsigned)nI1[c] <= hm ? nI1[c] : WF_NULL;
          nI2[c] = (unsigned)nI2[c] <= hm ? nI2[c] : WF_NULL;
          nD1[c] = (unsigned)nD1[c] <= hm ? nD1[c] : WF_NULL;
          nD2[c] = (unsigned)nD2[c] <= hm ? nD2[c] : WF_NULL;
          const int mis = (unsigned)nMis[c] <= hm ? nMis[c] : WF_NULL;
          nM[c] = max(max(max(nI1[c], nI2[c]), mis), max(nD1[c], nD2[c]));
          nM[c] = s <= s_last[c] ? nM[c] : WF_NULL;
        }
      }
    } else {
      // a column outside [-pl, tl], or one the score bound has cut off at this score, holds no cell
#pragma unroll
      for (int c = 0; c < C; ++c) nM[c] = s <= s_last[c] ? nM[c] : WF_NULL;
    }
    if (PIPE) {
      // ---- the pending row's rare paths, then it is final: into its line (the step before's class and visit), its antidiagonals into the maxima
      pk_rare2_fast(SRC, PR, pM, poP, poT, pmaxn, wlim, pext, lowdiv);
      int fin[C], mak = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        fin[c] = pM[c] + pext[c];  // (a cell that holds nothing probed at most 16 bases: NULL + 16 is as NULL as NULL)
        mak = max(mak, pM[c] >= 0 ? 2 * fin[c] + negk[c] : 0);  // (cells outside the core: far below zero)
      }
      switch (jj) {  // (compile time after unrolling: the step before's number within its body)
        case 1: line_take(std::integral_constant<int, UB>{}, fin); break;
        case 2: line_take(std::integral_constant<int, 1>{}, fin); break;
        case 3: line_take(std::integral_constant<int, 2>{}, fin); break;
        case 4: line_take(std::integral_constant<int, 3>{}, fin); break;
        case 5: line_take(std::integral_constant<int, 4>{}, fin); break;
        case 6: line_take(std::integral_constant<int, 5>{}, fin); break;
        case 7: line_take(std::integral_constant<int, 6>{}, fin); break;
        case 8: line_take(std::integral_constant<int, 7>{}, fin); break;
        case 9: line_take(std::integral_constant<int, 8>{}, fin); break;
        default: line_take(std::integral_constant<int, 9>{}, fin); break;
      }
      if (!FINE) (void)__hip_atomic_fetch_max(&s_run[tid], mak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_max_i32, nothing returned
      else {
        // (Tried in round 6 and taken out: four row steps and an LDS maximum from the four row ends instead of six DPP steps and a masked store --
        // ten instructions less in the instantiation C1 / C2 / C4 run every block of, and nothing in their device time: gpurun_out/r6y/ab4.log.)
        mak = wave_max63(mak);
        if (lane == 63) s_makr[(WAVE1 ? 0 : wv * MKS) + t - 1] = mak;
      }
      // ---- this step's row waits for its extension; its gap components are final
      if (__builtin_expect(t > t_stream, 0)) {  // stream the last rows of I/D of the core to the output snapshot
        stream_gap_rows(s, nI1, nI2, nD1, nD2);
      }
#pragma unroll
      for (int c = 0; c < C; ++c) { pM[c] = nM[c]; I1h[c][e1x] = nI1[c]; D1h[c][e1x] = nD1[c]; I2h[c] = nI2[c]; D2h[c] = nD2[c]; }
    } else {
    // ---- extension (pk_extend2): 16 bases of every cell at once from the LDS windows, longer runs in stages
    int ext[C], maxn[C];
    unsigned oP[C], oT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int m = nM[c];
      oP[c] = (unsigned)(m + cP[c]); oT[c] = (unsigned)(m + dT);
      maxn[c] = (FAST || m >= 0) ? (int)hmaxu[c] - m : 0;
    }
    WFM_TRACE_STAMP(1);  // recurrences issued
    if (FAST) pk_extend2_fast(SRC, nM, oP, oT, maxn, wlim, ext, lowdiv);
    else pk_extend2<false>(SRC, nM, oP, oT, maxn, ext, lowdiv);
#ifdef WFM_TILE_TRACE
    if (trace_on) asm volatile("" :: "v"(ext[0]), "v"(ext[1]));
#endif
    WFM_TRACE_STAMP(2);  // extension known
    int mak = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int m = nM[c];
      const int me = m + ext[c];
      const bool live = m >= 0;
      nM[c] = (FAST || live) ? me : WF_NULL;  // (FAST: a cell that holds nothing probed at most 16 bases: NULL + 16 is as NULL as NULL)
      mak = max(mak, live ? 2 * me + negk[c] : 0);  // (cells outside the core: far below zero)
    }
    if (P2) {
      // every row of the core is kept: five components into the job's P2 rows
      int32_t* prow = p2_arena + J.p2_off + J.koff2 + ((int64_t)(dir * 5) * P2K + (t - 1)) * J.w2;
      const int64_t cstride = (int64_t)P2K * J.w2;
      const bool w0 = incore[0] && s <= s_last[0], w1 = incore[1] && s <= s_last[1];
      if (w0 && w1) {
        st_pair(prow + C_M * cstride + k0, nM[0], nM[1]); st_pair(prow + C_I1 * cstride + k0, nI1[0], nI1[1]); st_pair(prow + C_I2 * cstride + k0, nI2[0], nI2[1]);
        st_pair(prow + C_D1 * cstride + k0, nD1[0], nD1[1]); st_pair(prow + C_D2 * cstride + k0, nD2[0], nD2[1]);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int k = k0 + c;
          if (c == 0 ? w0 : w1) {
            prow[C_M * cstride + k] = nM[c]; prow[C_I1 * cstride + k] = nI1[c]; prow[C_I2 * cstride + k] = nI2[c];
            prow[C_D1 * cstride + k] = nD1[c]; prow[C_D2 * cstride + k] = nD2[c];
          }
        }
      }
    }
    // stream the last H rows of I/D of the core to the output snapshot
    if (!P2 && __builtin_expect(t > t_stream, 0)) {
      stream_gap_rows(s, nI1, nI2, nD1, nD2);
    }
    // advance the delay line of this step's class only
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (!FAST) {
#pragma unroll
        for (int e = DEP - 1; e > 0; --e) Mh[c][cl][e] = Mh[c][cl][e - 1];
        Mh[c][cl][0] = nM[c];
#pragma unroll
        for (int d = E1 - 1; d > 0; --d) { I1h[c][d] = I1h[c][d - 1]; D1h[c][d] = D1h[c][d - 1]; }
        I1h[c][0] = nI1[c]; D1h[c][0] = nD1[c];
      } else {
        if (!vB) Mh[c][cl][DEP - 1] = nM[c];
        else {
          const int parked = Mh[c][cl][DEP - 1];
#pragma unroll
          for (int e = DEP - 1; e > 1; --e) Mh[c][cl][e] = Mh[c][cl][e - 2];
          Mh[c][cl][1] = parked;
          Mh[c][cl][0] = nM[c];
        }
        I1h[c][e1x] = nI1[c]; D1h[c][e1x] = nD1[c];
      }
      I2h[c] = nI2[c]; D2h[c] = nD2[c];
    }
    if (!P2) {  // (the rows of phase 2 are tested cell by cell: nobody reads maxima of theirs)
      if (FAST && !FINE) (void)__hip_atomic_fetch_max(&s_run[tid], mak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_max_i32, nothing returned
      else {
        mak = wave_max63(mak);
        if (FAST) { if (lane == 63) s_makr[(WAVE1 ? 0 : wv * MKS) + t] = mak; }
        else if (lane == 63 && mak > 0) {
          if (WAVE1) s_makr[t] = mak;
          else atomicMax(&s_makr[t], mak);
        }
      }
    }
    }
#undef M_S5
#undef M_S10
#undef M_S25
  }
  }
  if (PIPE) {
    // the last step's row is still pending: its extension, then into its line
    int pext[C], pmaxn[C], fin[C], mak = 0;
    unsigned poP[C], poT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { poP[c] = (unsigned)(pM[c] + cP[c]); poT[c] = (unsigned)(pM[c] + dT); pmaxn[c] = (int)hmaxu[c] - pM[c]; }
    pk_extend2_fast(SRC, pM, poP, poT, pmaxn, wlim, pext, lowdiv);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      fin[c] = pM[c] + pext[c];
      mak = max(mak, pM[c] >= 0 ? 2 * fin[c] + negk[c] : 0);
    }
    switch (Tn % UB) {
      case 1: line_take(std::integral_constant<int, 1>{}, fin); break;
      case 2: line_take(std::integral_constant<int, 2>{}, fin); break;
      case 3: line_take(std::integral_constant<int, 3>{}, fin); break;
      case 4: line_take(std::integral_constant<int, 4>{}, fin); break;
      case 5: line_take(std::integral_constant<int, 5>{}, fin); break;
      case 6: line_take(std::integral_constant<int, 6>{}, fin); break;
      case 7: line_take(std::integral_constant<int, 7>{}, fin); break;
      case 8: line_take(std::integral_constant<int, 8>{}, fin); break;
      case 9: line_take(std::integral_constant<int, 9>{}, fin); break;
      default: line_take(std::integral_constant<int, UB>{}, fin); break;
    }
    if (!FINE) (void)__hip_atomic_fetch_max(&s_run[tid], mak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else { mak = wave_max63(mak); if (lane == 63) s_makr[(WAVE1 ? 0 : wv * MKS) + Tn] = mak; }
  }
  WFM_TRACE_MARK(122);
  // ---- output snapshot: the newest H rows of M for the core ----
  if (P2) return;
  if (FAST) {
    // the classes whose row of the last, partial body is still parked: their line takes it now
    const int r10 = Tn % UB;
#pragma unroll
    for (int j0 = 1; j0 <= NCL; ++j0) {
      if (r10 >= j0 && r10 < j0 + NCL) {
        const int clx = j0 % NCL;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int parked = Mh[c][clx][DEP - 1];
#pragma unroll
          for (int e = DEP - 1; e > 0; --e) Mh[c][clx][e] = Mh[c][clx][e - 1];
          Mh[c][clx][0] = parked;
        }
      }
    }
  }
  const int s_end = s0 + Tn;
  if (Tn < H) {
    // a short last block: the I/D rows of scores <= s0 that the step kernel still looks at live in the input ring
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      if (!incore[c]) continue;
      for (int d = Tn; d < H; ++d) {
        const int sc = s_end - d;
        if (sc < 0 || k < rng2_lo(RG, sc) || k > rng2_hi(RG, sc)) continue;
        const int64_t ro = ((int64_t)(sc & RMASK)) * width + k;
        rout[(int64_t)C_I1 * RING * width + ro] = rin[(int64_t)C_I1 * RING * width + ro];
        rout[(int64_t)C_I2 * RING * width + ro] = rin[(int64_t)C_I2 * RING * width + ro];
        rout[(int64_t)C_D1 * RING * width + ro] = rin[(int64_t)C_D1 * RING * width + ro];
        rout[(int64_t)C_D2 * RING * width + ro] = rin[(int64_t)C_D2 * RING * width + ro];
      }
    }
  }
  auto write_rows = [&](auto TR) {
    constexpr int tr = decltype(TR)::value;  // T mod 5
    if (!incore[0] && !incore[1]) return;
#pragma unroll
    for (int d = 0; d < H; ++d) {
      constexpr int dummy = 0; (void)dummy;
      const int r = ((tr - d) % NCL + NCL) % NCL;        // class of row s_end - d
      const int back = ((tr - r) % NCL + NCL) % NCL;     // s_end - (newest score of class r)
      const int e = (d - back) / NCL;
      const int sc = s_end - d;
      const int lo = rng2_lo(RG, sc), hi = rng2_hi(RG, sc);
      const bool w0 = incore[0] && sc >= 0 && k0 >= lo && k0 <= hi, w1 = incore[1] && sc >= 0 && k0 + 1 >= lo && k0 + 1 <= hi;
      int32_t* dst = rout + ((int64_t)(C_M * RING + (sc & RMASK))) * width + k0;
      // (a lane's two diagonals as one 8-byte store wherever both are the core's and inside the row: all lanes but a handful at the edges)
      if (w0 && w1) st_pair(dst, Mh[0][r][e], Mh[1][r][e]);
      else {
        if (w0) dst[0] = Mh[0][r][e];
        if (w1) dst[1] = Mh[1][r][e];
      }
    }
  };
  switch (Tn % NCL) {
    case 0: write_rows(std::integral_constant<int, 0>{}); break;
    case 1: write_rows(std::integral_constant<int, 1>{}); break;
    case 2: write_rows(std::integral_constant<int, 2>{}); break;
    case 3: write_rows(std::integral_constant<int, 3>{}); break;
    default: write_rows(std::integral_constant<int, 4>{}); break;
  }
  int32_t* mk = mak_out + ((int64_t)tk.job * 2 + dir) * T;
  WFM_TRACE_MARK(123);
  if (!FINE) {
    // one maximum for the whole block, in the block's last slot: the advance kernel's prefix maxima make of it what the per-score form would
    // have left at the block's end -- enough to say whether the directions met in this block (they are run again with per-score maxima then)
    if (J.mode == 0) {
      const int makrun = wave_max63(s_run[tid]);
      if (lane == 63 && makrun > 0) atomicMax(&mk[T - 1], makrun);
    }
    return;
  }
  __syncthreads();
  for (int t = 1 + tid; t <= T; t += NT) {
    int v = s_makr[t];
    if (FAST && !WAVE1) for (int w = 1; w < nw; ++w) v = max(v, s_makr[w * MKS + t]);
    if (v > 0) atomicMax(&mk[t - 1], v);
  }
}

static bool tile_fast() {  // (read per launch: the tests switch forms inside one process)
  const char* e = getenv("WFM_TILE_FAST");
  return !(e && atoi(e) == 0);
}
bool tile2_coarse_maxima() {  // (read per launch, like tile_fast)
  const char* e = getenv("WFM_TILE_COARSE");
  return tile_fast() && !(e && atoi(e) == 0);
}
// variants: bit 0 -- the instantiation without per-score maxima, bit 1 -- the one with them (wfa_tile2_kernel, FINE); the FINE one alone takes
// every tile (per-score maxima are a superset of what any job's block needs).  WFM_TILE_COARSE=0 / the round-4 form: one launch, maxima always
void launch_tile2(const uint32_t* pk, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int32_t* mak, int ntasks, int threads, int T,
                  int variants, hipStream_t st) {
  const size_t lds1 = (size_t)(T + 1) * 4;
  if (tile_fast()) {
    const int coarse = tile2_coarse_maxima() ? 1 : 0;
    if (!coarse) variants = 2;
    // (WFM_TILE_LDS_PAD: bytes of LDS nobody uses, to take workgroups off a CU -- the occupancy experiment of DESIGN section 5, round 6)
    const size_t pad = getenv("WFM_TILE_LDS_PAD") ? (size_t)atoi(getenv("WFM_TILE_LDS_PAD")) : 0;
    if (variants & 1) {
      if (threads <= 64) hipLaunchKernelGGL((wfa_tile2_kernel<64, false, true, false>), dim3(ntasks), dim3(64), 0, st, pk, ring, jobs, tasks, mak, T, (int32_t*)nullptr, coarse);
      else hipLaunchKernelGGL((wfa_tile2_kernel<1024, false, true, false>), dim3(ntasks), dim3(threads), pad, st, pk, ring, jobs, tasks, mak, T, (int32_t*)nullptr, coarse);
    }
    if (variants & 2) {
      const int cf = coarse | (variants == 2 ? 2 : 0);
      if (threads <= 64) hipLaunchKernelGGL((wfa_tile2_kernel<64, false, true, true>), dim3(ntasks), dim3(64), lds1, st, pk, ring, jobs, tasks, mak, T, (int32_t*)nullptr, cf);
      else hipLaunchKernelGGL((wfa_tile2_kernel<1024, false, true, true>), dim3(ntasks), dim3(threads), lds1 * (size_t)(threads / 64), st, pk, ring, jobs, tasks, mak, T, (int32_t*)nullptr, cf);
    }
  } else {
    if (threads <= 64) hipLaunchKernelGGL((wfa_tile2_kernel<64, false, false, true>), dim3(ntasks), dim3(64), lds1, st, pk, ring, jobs, tasks, mak, T, (int32_t*)nullptr, 0);
    else hipLaunchKernelGGL((wfa_tile2_kernel<1024, false, false, true>), dim3(ntasks), dim3(threads), lds1, st, pk, ring, jobs, tasks, mak, T, (int32_t*)nullptr, 0);
  }
}
void launch_tile2_p2(const uint32_t* pk, int32_t* ring, const TileJob* jobs, const TileTask* tasks, int ntasks, int threads, int32_t* p2, hipStream_t st) {
  const size_t lds1 = (size_t)(P2K + 1) * 4;
  if (tile_fast()) {
    if (threads <= 64) hipLaunchKernelGGL((wfa_tile2_kernel<64, true, true, false>), dim3(ntasks), dim3(64), 0, st, pk, ring, jobs, tasks, (int32_t*)nullptr, P2K, p2, 0);
    else hipLaunchKernelGGL((wfa_tile2_kernel<1024, true, true, false>), dim3(ntasks), dim3(threads), 0, st, pk, ring, jobs, tasks, (int32_t*)nullptr, P2K, p2, 0);
  } else {
    if (threads <= 64) hipLaunchKernelGGL((wfa_tile2_kernel<64, true, false, true>), dim3(ntasks), dim3(64), lds1, st, pk, ring, jobs, tasks, (int32_t*)nullptr, P2K, p2, 0);
    else hipLaunchKernelGGL((wfa_tile2_kernel<1024, true, false, true>), dim3(ntasks), dim3(threads), lds1, st, pk, ring, jobs, tasks, (int32_t*)nullptr, P2K, p2, 0);
  }
}

// ---------------------------------------------------------------------------
// Leaves and patches on registers (default penalties, packed sequences): wfa_base_kernel's contract
// ---------------------------------------------------------------------------
// r32::wfa_base_kernel keeps its rows in a global-memory ring: a score step reads nine values per cell from L2, extends on
// bytes from L2 and writes seven values back, with a workgroup barrier in between -- 3 to 5 microseconds per step, a leaf is
// 250 steps deep and a patch that overflowed its first budget 2000.  Here a workgroup holds the whole row of its job in the
// tile kernel's register delay lines (two diagonals per lane, up to 2048 diagonals), takes its neighbours by wave shifts and
// the LDS mailbox, extends on the packed windows (a leaf's or a patch's sequences fit them whole), and writes only what the
// backtrace reads: per cell the offset before the extension and one byte of decisions, the same bits decided the same way
// (extension wins ties over opening: BT_*_EXT; source of M on equal offsets: mismatch > D2 > D1 > I2 > I1).  The row ranges
// the result counts as cells are the ones the ring kernel keeps (per row the union of its sources' ranges), and the
// backtrace is its backtrace.  Jobs with other penalties, an N, wider rows or longer sequences stay with the ring kernel.
struct RleWriter2 {
  uint32_t* base;  // entries are written at base[-1], base[-2], ...
  int n;
  int cur_op;
  uint32_t cur_len;
  bool writes;
  __device__ void push(int op, int len) {
    if (len <= 0) return;
    if (op == cur_op) { cur_len += (uint32_t)len; return; }
    flush();
    cur_op = op; cur_len = (uint32_t)len;
  }
  __device__ void flush() {
    if (cur_len) { ++n; if (writes) base[-n] = (cur_len << 2) | (uint32_t)cur_op; }
    cur_len = 0; cur_op = -1;
  }
};

// wavefront_backtrace_affine over the rows a forward pass has left (pre: the offset of every M cell before its extension, bt: its decision byte; both
// addressed [score][diagonal] with the job's width): every lane of the wave with the same state, lane 0 writes; a run of gap cells is read 64 decision
// bytes at a time (r32::wfa_base_kernel).  Returns the number of runs written below rle[J.rle_end].
__device__ __forceinline__ int base2_walk(const BaseJob& J, const int32_t* __restrict__ pre_base, const uint8_t* __restrict__ bt_base, uint32_t* __restrict__ rle,
                                          int s, int k_from, int off_from, int lane) {
  constexpr int PX = 5, PO1 = 8, PE1 = 2, PO2 = 24, PE2 = 1;
  const int pl = J.pl, tl = J.tl;
  const int64_t width = J.width;
  RleWriter2 w; w.base = rle + J.rle_end; w.n = 0; w.cur_op = -1; w.cur_len = 0; w.writes = lane == 0;
  int comp = J.endsfree ? C_M : J.comp_end;
  int k = k_from;
  int off = off_from;
  int sc = s;
  int h = off, v = off - k;
  if (comp == C_M) {
    if (v < pl) w.push(OP_D, pl - v);
    if (h < tl) w.push(OP_I, tl - h);
  }
  while (v > 0 && h > 0 && sc > 0) {
    if (comp != C_M) {
      const bool ins = comp == C_I1 || comp == C_I2;
      const int e = (comp == C_I1 || comp == C_D1) ? PE1 : PE2, o = (comp == C_I1 || comp == C_D1) ? PO1 : PO2;
      const unsigned mask = comp == C_I1 ? BT_I1_EXT : (comp == C_I2 ? BT_I2_EXT : (comp == C_D1 ? BT_D1_EXT : BT_D2_EXT));
      const int scj = sc - lane * e, kj = ins ? k - lane : k + lane;
      const bool alive = scj > 0 && (ins ? h - lane > 0 : v - lane > 0);
      const unsigned bj = alive ? bt_base[(int64_t)scj * width + kj] : 0u;
      const unsigned long long stop = __ballot(!(alive && (bj & mask)));
      const int j0 = stop ? (int)__builtin_ctzll(stop) : 64;  // cells 0 .. j0-1 continue the gap
      if (j0 > 0) {
        w.push(ins ? OP_I : OP_D, j0);
        sc -= j0 * e;
        if (ins) { k -= j0; off -= j0; } else k += j0;
        v = off - k; h = off;
      }
      if (j0 < 64) {
        if (!(v > 0 && h > 0 && sc > 0)) break;   // the walk ends inside the gap
        sc -= o + e; comp = C_M;                  // the cell that opened the gap
        w.push(ins ? OP_I : OP_D, 1);
        if (ins) { --k; --off; } else ++k;
        v = off - k; h = off;
      }
      continue;
    }
    // (round 6, as in the ring kernel's walk -- wfa_generic_inc.h: a run of mismatches stays on its diagonal, PX scores apart: the lanes read
    // the next 64 cells of that line at once, the walk goes through them from registers while each one's source is the mismatch)
    const int scj = sc - lane * PX;
    const unsigned bj = scj > 0 ? (unsigned)bt_base[(int64_t)scj * width + k] : 0u;
    const int pj = scj > 0 ? pre_base[(int64_t)scj * width + k] : 0;
    bool stop = false;
    for (int j = 0; j < 64; ++j) {
      const unsigned b = (unsigned)rdl((int)bj, j);
      const int pre = rdl(pj, j);
      w.push(OP_M, off - pre);
      off = pre; v = off - k; h = off;
      if (v <= 0 || h <= 0) { stop = true; break; }
      const unsigned src = b & 7u;
      if (src == C_M) {
        sc -= PX; comp = C_M; w.push(OP_X, 1); --off;
        v = off - k; h = off;
        if (!(v > 0 && h > 0 && sc > 0)) break;
        continue;
      }
      if (src == C_I1) { if (b & BT_I1_EXT) { sc -= PE1; comp = C_I1; } else { sc -= PO1 + PE1; comp = C_M; } w.push(OP_I, 1); --k; --off; }
      else if (src == C_I2) { if (b & BT_I2_EXT) { sc -= PE2; comp = C_I2; } else { sc -= PO2 + PE2; comp = C_M; } w.push(OP_I, 1); --k; --off; }
      else if (src == C_D1) { if (b & BT_D1_EXT) { sc -= PE1; comp = C_D1; } else { sc -= PO1 + PE1; comp = C_M; } w.push(OP_D, 1); ++k; }
      else { if (b & BT_D2_EXT) { sc -= PE2; comp = C_D2; } else { sc -= PO2 + PE2; comp = C_M; } w.push(OP_D, 1); ++k; }
      v = off - k; h = off;
      break;
    }
    if (stop) break;
  }
  if (comp == C_M && v > 0 && h > 0) { const int nm = min(v, h); w.push(OP_M, nm); v -= nm; h -= nm; }
  if (v > 0) w.push(OP_D, v);
  if (h > 0) w.push(OP_I, h);
  w.flush();
  return w.n;
}

template <int NTMAX>
__global__ __launch_bounds__(NTMAX) void wfa_base2_kernel(const uint32_t* __restrict__ pk, int32_t* __restrict__ arena32, uint8_t* __restrict__ arena8,
                                                        uint32_t* __restrict__ rle, const BaseJob* __restrict__ jobs, BaseResult* __restrict__ results) {
  constexpr int C = 2, NCL = 5, DEP = 6, E1 = 2;
  constexpr int PX = 5, PO1 = 8, PE1 = 2, PO2 = 24, PE2 = 1;  // the penalties this form is built for (the host checks)
  constexpr bool WAVE1 = NTMAX == 64;
  const BaseJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (a launch is sized for its widest job: the waves this one's rows do not reach end before the first barrier, see wfa_tile2_kernel)
  const int nw = J.type != 0 ? 1 : min((int)(blockDim.x >> 6), (int)((J.width - 1) / (64 * 2)) + 1), NT = nw * 64;
  if (wv >= nw) return;
  const long long t_begin = wall_clock64();
  if (J.type != 0) {  // trivial: all-D or all-I (wavefront_bialign_alignment trivial cases)
    if (tid == 0) {
      BaseResult r; r.status = 0; r.cells = 0; r.nruns = 0; r.score = 0; r.pad_ = 0;
      const int len = J.type == 1 ? J.pl : J.tl;
      if (len > 0) { rle[J.rle_end - 1] = ((uint32_t)len << 2) | (uint32_t)(J.type == 1 ? OP_D : OP_I); r.nruns = 1; }
      results[blockIdx.x] = r;
    }
    return;
  }
  __shared__ int s_edge[2][WAVE1 ? 1 : 16][2][4];
  __shared__ __attribute__((aligned(16))) uint32_t s_winP[PK_WIN_DW + PK_SLACK_DW], s_winT[PK_WIN_DW + PK_SLACK_DW];
  __shared__ int s_done, s_endk, s_endoff;
  const int pl = J.pl, tl = J.tl, kmin = J.kmin, kmax = J.kmin + J.width - 1;
  const int64_t width = J.width;
  int32_t* pre_base = arena32 + J.pre_off - kmin;
  uint8_t* bt_base = arena8 + J.bt_off - kmin;
  const int k_end = tl - pl;
  // ---- the sequences whole in the windows (the host sends only jobs whose sequences fit)
  const int64_t oriP = J.p_off & ~(int64_t)15, oriT = J.t_off & ~(int64_t)15;
  const int dP = (int)(J.p_off - oriP), dT = (int)(J.t_off - oriT);
  PkSrc SRC;
  SRC.lP = (lds_words)s_winP; SRC.lT = (lds_words)s_winT;
  SRC.gP = (glb_words)pk + (oriP >> 4); SRC.gT = (glb_words)pk + (oriT >> 4);
  {
    const int nP = min(PK_WIN_DW + PK_SLACK_DW, (pl + dP + 15) / 16 + 8), nT = min(PK_WIN_DW + PK_SLACK_DW, (tl + dT + 15) / 16 + 8);
    for (int i = tid; i < nP; i += NT) s_winP[i] = SRC.gP[i];
    for (int i = tid; i < nT; i += NT) s_winT[i] = SRC.gT[i];
  }
  if (tid == 0) { s_done = 0; s_endk = INT32_MAX; s_endoff = 0; }
  for (int i = tid; i < (int)(sizeof(s_edge) / sizeof(int)); i += NT) ((int*)s_edge)[i] = WF_NULL;  // (a wave that has no cell yet publishes nothing)
  __syncthreads();

  const int k0 = kmin + tid * C;
  const int kw_lo = kmin + (tid & ~63) * C, kw_hi = kw_lo + 64 * C - 1;  // the diagonals of this thread's wave
  int Mh[C][NCL][DEP];
  int I1h[C][E1], D1h[C][E1], I2h[C], D2h[C];
  unsigned hmaxu[C];
  int cP[C], mcur[C], wlim[C];
  bool colok[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = k0 + c;
#pragma unroll
    for (int r = 0; r < NCL; ++r)
#pragma unroll
      for (int e = 0; e < DEP; ++e) Mh[c][r][e] = WF_NULL;
#pragma unroll
    for (int d = 0; d < E1; ++d) { I1h[c][d] = WF_NULL; D1h[c][d] = WF_NULL; }
    I2h[c] = WF_NULL; D2h[c] = WF_NULL;
    colok[c] = k >= -pl && k <= tl && k <= kmax;
    hmaxu[c] = colok[c] ? (unsigned)min(tl, pl + k) : 0u;
    cP[c] = dP - k;
    wlim[c] = PK_WIN_BASES - 1 - max(cP[c], dT);
    mcur[c] = WF_NULL;
  }
  auto end_checks = [&](int c, int k, int m_ext, int ins1, int ins2, int del1, int del2) {
    if (J.endsfree) {
      if (m_ext >= 0) {
        const int h = m_ext, v = m_ext - k;
        if ((h >= tl && pl - v <= J.pef) || (v >= pl && tl - h <= J.tef)) atomicMin(&s_endk, k);
      }
    } else if (k == k_end) {
      const int ev = J.comp_end == C_M ? m_ext : (J.comp_end == C_I1 ? ins1 : (J.comp_end == C_I2 ? ins2 : (J.comp_end == C_D1 ? del1 : del2)));
      if (ev >= tl) s_done = 1;
    }
    (void)c;
  };
  // ---- row 0
  int lo0, hi0;
  if (J.endsfree) { lo0 = max(-J.pbf, kmin); hi0 = min(J.tbf, kmax); }
  else { lo0 = 0; hi0 = 0; }
  {
    int m0[C], ext[C], maxn[C];
    unsigned oP[C], oT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      const bool on = k >= lo0 && k <= hi0;
      int m = WF_NULL;
      if (on) {
        if (J.endsfree) m = k > 0 ? k : 0;
        else {
          if (J.comp_begin == C_M) m = 0;
          I1h[c][0] = J.comp_begin == C_I1 ? 0 : WF_NULL;
          I2h[c] = J.comp_begin == C_I2 ? 0 : WF_NULL;
          D1h[c][0] = J.comp_begin == C_D1 ? 0 : WF_NULL;
          D2h[c] = J.comp_begin == C_D2 ? 0 : WF_NULL;
        }
      }
      m0[c] = m;
      oP[c] = (unsigned)(m + cP[c]); oT[c] = (unsigned)(m + dT);
      maxn[c] = m >= 0 ? min(pl - (m - k), tl - m) : 0;
    }
    pk_extend2(SRC, m0, oP, oT, maxn, ext);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      if (k < lo0 || k > hi0) continue;
      pre_base[k] = m0[c]; bt_base[k] = 0;
      int m = m0[c];
      if (m >= 0) m += ext[c];
      if (J.endsfree) end_checks(c, k, m, WF_NULL, WF_NULL, WF_NULL, WF_NULL);
      else if (k == k_end && J.comp_end == C_M && m >= tl) s_done = 1;
      Mh[c][0][0] = m;
      mcur[c] = m;
    }
  }
  uint64_t cells = (uint64_t)(hi0 - lo0 + 1);
  int s = 0, status = 0;
  bool done = false;
  // ---- the steps: score s = tb + jj, class jj % 5 at compile time
  for (int tb = 0; !done && status == 0; tb += NCL) {
#pragma unroll
  for (int jj = 1; jj <= NCL; ++jj) {
    const int cl = jj % NCL;
    const int sn = tb + jj;  // the score this step computes
    int lM10, lM25, lI1, lI2, rM10, rM25, rD1, rD2;
    const int par = sn & 1;
    if (!WAVE1) {
      if (lane == 63) { int* e = s_edge[par][wv][0]; e[0] = Mh[C - 1][cl][1]; e[1] = Mh[C - 1][cl][4]; e[2] = I1h[C - 1][E1 - 1]; e[3] = I2h[C - 1]; }
      if (lane == 0)  { int* e = s_edge[par][wv][1]; e[0] = Mh[0][cl][1];     e[1] = Mh[0][cl][4];     e[2] = D1h[0][E1 - 1];     e[3] = D2h[0]; }
    }
    lds_barrier();  // the previous step's end checks and row range are visible; the edges of this one are published (the rows of pre / bt are read after the loop's end)
    done = J.endsfree ? (s_endk != INT32_MAX) : (s_done != 0);
    if (done) break;
    if (sn > J.smax) { status = WFM_DEV_OVERFLOW; break; }
    s = sn;
    // The row's range as the ring kernel keeps it -- the union of its sources' ranges, the I / D sources reaching one diagonal
    // further -- in closed form: row s - 1 is among the sources (e2 = 1) and holds every older row, so a row is its predecessor
    // and one diagonal more on either side, clipped to the problem and to the job's columns
    const int lo = max(lo0 - s, max(-pl, kmin)), hi = min(hi0 + s, min(tl, kmax));
    const bool valid = lo <= hi;
    if (valid) cells += (uint64_t)(hi - lo + 1);
    // A wave whose diagonals lie outside the row (and one column around it) has nothing to do: rows only grow (a row holds its
    // predecessor's range and one diagonal more on either side), so everything the wave holds is NULL and stays NULL, and what
    // it would publish for its neighbours is what it published before.  The budget of a leaf is its exact score and the rows
    // are laid out for the budget: on average a third of the waves of a launch have cells at a given step.
    if (!valid || kw_hi < lo - 1 || kw_lo > hi + 1) continue;
    lM10 = from_prev_lane(Mh[C - 1][cl][1]); lM25 = from_prev_lane(Mh[C - 1][cl][4]);
    lI1 = from_prev_lane(I1h[C - 1][E1 - 1]); lI2 = from_prev_lane(I2h[C - 1]);
    rM10 = from_next_lane(Mh[0][cl][1]); rM25 = from_next_lane(Mh[0][cl][4]);
    rD1 = from_next_lane(D1h[0][E1 - 1]); rD2 = from_next_lane(D2h[0]);
    if (!WAVE1) {
      if (lane == 0 && wv > 0) { const int* e = s_edge[par][wv - 1][0]; lM10 = e[0]; lM25 = e[1]; lI1 = e[2]; lI2 = e[3]; }
      if (lane == 63 && wv + 1 < nw) { const int* e = s_edge[par][wv + 1][1]; rM10 = e[0]; rM25 = e[1]; rD1 = e[2]; rD2 = e[3]; }
    }
    int nM[C], nI1[C], nI2[C], nD1[C], nD2[C], preM[C];
    unsigned btb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      const int a10 = c == 0 ? lM10 : Mh[c - 1][cl][1], b10 = c == C - 1 ? rM10 : Mh[c + 1][cl][1];
      const int a25 = c == 0 ? lM25 : Mh[c - 1][cl][4], b25 = c == C - 1 ? rM25 : Mh[c + 1][cl][4];
      const int i1 = c == 0 ? lI1 : I1h[c - 1][E1 - 1], d1 = c == C - 1 ? rD1 : D1h[c + 1][E1 - 1];
      const int i2 = c == 0 ? lI2 : I2h[c - 1], d2 = c == C - 1 ? rD2 : D2h[c + 1];
      const int mx = Mh[c][cl][0];
      unsigned bits = 0;
      // ext wins ties (WFA2-lib: ext type > open type; piggyback: ext >= open)
      if (i1 >= a10) bits |= BT_I1_EXT;
      if (i2 >= a25) bits |= BT_I2_EXT;
      if (d1 >= b10) bits |= BT_D1_EXT;
      if (d2 >= b25) bits |= BT_D2_EXT;
      const unsigned hm = hmaxu[c];
      int ins1 = max(a10, i1) + 1, ins2 = max(a25, i2) + 1, del1 = max(b10, d1), del2 = max(b25, d2), mis = mx + 1;
      ins1 = (unsigned)ins1 <= hm ? ins1 : WF_NULL;
      ins2 = (unsigned)ins2 <= hm ? ins2 : WF_NULL;
      del1 = (unsigned)del1 <= hm ? del1 : WF_NULL;
      del2 = (unsigned)del2 <= hm ? del2 : WF_NULL;
      mis = (unsigned)mis <= hm ? mis : WF_NULL;
      // M source priority on equal offsets: mismatch > D2 > D1 > I2 > I1
      int m = ins1; unsigned src = C_I1;
      if (ins2 >= m) { m = ins2; src = C_I2; }
      if (del1 >= m) { m = del1; src = C_D1; }
      if (del2 >= m) { m = del2; src = C_D2; }
      if (mis >= m)  { m = mis;  src = C_M; }
      const bool on = valid && k >= lo && k <= hi;  // (outside the row's range the ring kernel computes nothing: nothing is kept there)
      nI1[c] = on ? ins1 : WF_NULL; nI2[c] = on ? ins2 : WF_NULL; nD1[c] = on ? del1 : WF_NULL; nD2[c] = on ? del2 : WF_NULL;
      nM[c] = (on && colok[c]) ? m : WF_NULL;
      preM[c] = nM[c];
      btb[c] = bits | src;
      (void)k;
    }
    int ext[C], maxn[C];
    unsigned oP[C], oT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int m = nM[c];
      oP[c] = (unsigned)(m + cP[c]); oT[c] = (unsigned)(m + dT);
      maxn[c] = (int)hmaxu[c] - m;  // (a cell that holds nothing: whatever -- pk_extend2_fast looks at m first)
    }
    // (round 6: the tile kernel's lean form of the staged extension -- masks kept as masks, the rare stages out of line -- and the end tests only
    // where a cell has reached the far edge of its diagonal, h = tl or v = pl <=> its offset is hmaxu: the step's common path went from the better
    // part of 940 instructions to 357)
    pk_extend2_fast(SRC, nM, oP, oT, maxn, wlim, ext, false);
    int32_t* pre = pre_base + (int64_t)s * width;
    uint8_t* bt = bt_base + (int64_t)s * width;
    bool reached = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      const bool on = valid && k >= lo && k <= hi;
      if (on) {
        pre[k] = preM[c];
        bt[k] = (uint8_t)btb[c];
        int m = nM[c];
        if (m >= 0) m += ext[c];
        nM[c] = m;
        reached |= !J.endsfree || m >= (int)hmaxu[c];
      }
    }
    if (__builtin_expect(__any(reached), 0)) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int k = k0 + c;
        if (valid && k >= lo && k <= hi) end_checks(c, k, nM[c], nI1[c], nI2[c], nD1[c], nD2[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      mcur[c] = nM[c];
#pragma unroll
      for (int e = DEP - 1; e > 0; --e) Mh[c][cl][e] = Mh[c][cl][e - 1];
      Mh[c][cl][0] = nM[c];
#pragma unroll
      for (int d = E1 - 1; d > 0; --d) { I1h[c][d] = I1h[c][d - 1]; D1h[c][d] = D1h[c][d - 1]; }
      I1h[c][0] = nI1[c]; D1h[c][0] = nD1[c];
      I2h[c] = nI2[c]; D2h[c] = nD2[c];
    }
  }
  }
  // ---- the ends-free walk starts from the offset its end cell holds
  if (status == 0 && J.endsfree) {
#pragma unroll
    for (int c = 0; c < C; ++c) if (k0 + c == s_endk) s_endoff = mcur[c];
  }
  __syncthreads();
  // ---- backtrace (wavefront_backtrace_affine): the first wave, every lane with the same state, lane 0 writes; a run of gap
  // cells is read 64 decision bytes at a time (r32::wfa_base_kernel)
  if (tid < 64) {
    const long long t_fwd = wall_clock64();
    BaseResult r; r.status = status; r.score = s; r.nruns = 0; r.cells = cells; r.pad_ = 0;
    if (status == 0) r.nruns = base2_walk(J, pre_base, bt_base, rle, s, J.endsfree ? s_endk : k_end, J.endsfree ? s_endoff : tl, lane);
    // diagnostics (WFM_DEBUG=2): microseconds of the forward pass and of the walk back, 16 bits each
    r.pad_ = (int32_t)((min((t_fwd - t_begin) / 100, 65535ll) << 16) | min((wall_clock64() - t_fwd) / 100, 65535ll));
    if (lane == 0) results[blockIdx.x] = r;
  }
}

// ---------------------------------------------------------------------------
// Base jobs wider than a workgroup's registers: wfa_base2_kernel's step on tiles (Base2TJob in wfa_device.h)
// ---------------------------------------------------------------------------
// One workgroup = one tile of one job for one block of T scores (T a multiple of 5: a block begins with class 1).  Everything a cell computes and
// everything that is written about it is wfa_base2_kernel's -- same recurrences, same decision bits, same range of a row, same end tests -- but rows of
// pre / bt and the end tests are the core's alone: the halo's cells are right only as far from the tile's edge as the block is old.
template <int NTMAX>
__global__ __launch_bounds__(NTMAX) void wfa_base2t_kernel(const uint32_t* __restrict__ pk, int32_t* __restrict__ arena32, uint8_t* __restrict__ arena8,
                                                         const Base2TJob* __restrict__ jobs, const Base2TTask* __restrict__ tasks,
                                                         unsigned long long* __restrict__ tile_key, int32_t* __restrict__ tile_off, int T) {
  constexpr int C = 2, NCL = 5, DEP = 6, E1 = 2, H = 26;
  constexpr int NW = NTMAX / 64;
  const Base2TTask tk = tasks[blockIdx.x];
  const Base2TJob JT = jobs[tk.job];
  if (JT.done) return;
  const BaseJob& J = JT.b;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ int s_edge[2][NW][2][4];
  __shared__ __attribute__((aligned(16))) uint32_t s_winP[PK_WIN_DW + PK_SLACK_DW], s_winT[PK_WIN_DW + PK_SLACK_DW];
  __shared__ int s_done, s_endk, s_endoff;
  const int pl = J.pl, tl = J.tl, kmin = J.kmin, kmax = J.kmin + J.width - 1;
  const int64_t width = J.width;
  int32_t* pre_base = arena32 + J.pre_off - kmin;
  uint8_t* bt_base = arena8 + J.bt_off - kmin;
  const int32_t* snap_in = arena32 + JT.snap_in - kmin;
  int32_t* snap_out = arena32 + JT.snap_out - kmin;
  const int k_end = tl - pl;
  const int s0 = JT.s0;
  const int core_lo = kmin + tk.tile * JT.core, core_hi = min(kmax, core_lo + JT.core - 1);
  const int64_t oriP = J.p_off & ~(int64_t)15, oriT = J.t_off & ~(int64_t)15;
  const int dP = (int)(J.p_off - oriP), dT = (int)(J.t_off - oriT);
  PkSrc SRC;
  SRC.lP = (lds_words)s_winP; SRC.lT = (lds_words)s_winT;
  SRC.gP = (glb_words)pk + (oriP >> 4); SRC.gT = (glb_words)pk + (oriT >> 4);
  {
    const int nP = min(PK_WIN_DW + PK_SLACK_DW, (pl + dP + 15) / 16 + 8), nT = min(PK_WIN_DW + PK_SLACK_DW, (tl + dT + 15) / 16 + 8);
    for (int i = tid; i < nP; i += NTMAX) s_winP[i] = SRC.gP[i];
    for (int i = tid; i < nT; i += NTMAX) s_winT[i] = SRC.gT[i];
  }
  if (tid == 0) { s_done = 0; s_endk = INT32_MAX; s_endoff = 0; }
  for (int i = tid; i < (int)(sizeof(s_edge) / sizeof(int)); i += NTMAX) ((int*)s_edge)[i] = WF_NULL;
  __syncthreads();

  const int k0 = core_lo - T + tid * C;  // (the host sizes the core: core + 2 T <= NTMAX * C)
  const int kw_lo = core_lo - T + (tid & ~63) * C, kw_hi = kw_lo + 64 * C - 1;  // the diagonals of this thread's wave
  int Mh[C][NCL][DEP];
  int I1h[C][E1], D1h[C][E1], I2h[C], D2h[C];
  unsigned hmaxu[C];
  int cP[C], mcur[C], wlim[C];
  bool colok[C], incore[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = k0 + c;
#pragma unroll
    for (int r = 0; r < NCL; ++r)
#pragma unroll
      for (int e = 0; e < DEP; ++e) Mh[c][r][e] = WF_NULL;
#pragma unroll
    for (int d = 0; d < E1; ++d) { I1h[c][d] = WF_NULL; D1h[c][d] = WF_NULL; }
    I2h[c] = WF_NULL; D2h[c] = WF_NULL;
    colok[c] = k >= -pl && k <= tl && k >= kmin && k <= kmax;
    incore[c] = k >= core_lo && k <= core_hi;
    hmaxu[c] = colok[c] ? (unsigned)min(tl, pl + k) : 0u;
    cP[c] = dP - k;
    wlim[c] = PK_WIN_BASES - 1 - max(cP[c], dT);
    mcur[c] = WF_NULL;
  }
  auto end_checks = [&](int k, int m_ext, int ins1, int ins2, int del1, int del2) {
    if (J.endsfree) {
      if (m_ext >= 0) {
        const int h = m_ext, v = m_ext - k;
        if ((h >= tl && pl - v <= J.pef) || (v >= pl && tl - h <= J.tef)) atomicMin(&s_endk, k);
      }
    } else if (k == k_end) {
      const int ev = J.comp_end == C_M ? m_ext : (J.comp_end == C_I1 ? ins1 : (J.comp_end == C_I2 ? ins2 : (J.comp_end == C_D1 ? del1 : del2)));
      if (ev >= tl) s_done = 1;
    }
  };
  int lo0, hi0;
  if (J.endsfree) { lo0 = max(-J.pbf, kmin); hi0 = min(J.tbf, kmax); }
  else { lo0 = 0; hi0 = 0; }
  if (s0 == 0) {
    // ---- row 0 (no cell of it has a neighbour to wait for: the halo's are right)
    int m0[C], ext[C], maxn[C];
    unsigned oP[C], oT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      const bool on = k >= lo0 && k <= hi0;
      int m = WF_NULL;
      if (on) {
        if (J.endsfree) m = k > 0 ? k : 0;
        else {
          if (J.comp_begin == C_M) m = 0;
          I1h[c][0] = J.comp_begin == C_I1 ? 0 : WF_NULL;
          I2h[c] = J.comp_begin == C_I2 ? 0 : WF_NULL;
          D1h[c][0] = J.comp_begin == C_D1 ? 0 : WF_NULL;
          D2h[c] = J.comp_begin == C_D2 ? 0 : WF_NULL;
        }
      }
      m0[c] = m;
      oP[c] = (unsigned)(m + cP[c]); oT[c] = (unsigned)(m + dT);
      maxn[c] = m >= 0 ? min(pl - (m - k), tl - m) : 0;
    }
    pk_extend2(SRC, m0, oP, oT, maxn, ext);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      if (k < lo0 || k > hi0) continue;
      if (incore[c]) { pre_base[k] = m0[c]; bt_base[k] = 0; }
      int m = m0[c];
      if (m >= 0) m += ext[c];
      if (incore[c]) {
        if (J.endsfree) end_checks(k, m, WF_NULL, WF_NULL, WF_NULL, WF_NULL);
        else if (k == k_end && J.comp_end == C_M && m >= tl) s_done = 1;
      }
      Mh[c][0][0] = m;
      mcur[c] = m;
    }
  } else {
    // ---- the snapshot of the block before: rows 0 .. 25 = M of scores s0, s0 - 1, ..; 26 / 27 = I1 of s0 / s0 - 1; 28 / 29 = D1; 30 = I2; 31 = D2.
    // Every column of the job lies in one tile's core and was written there, NULL where the row had no cell
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      if (k < kmin || k > kmax) continue;
#pragma unroll
      for (int d = 0; d < H; ++d) Mh[c][(NCL - d % NCL) % NCL][d / NCL] = snap_in[(int64_t)d * width + k];
#pragma unroll
      for (int d = 0; d < E1; ++d) { I1h[c][d] = snap_in[(int64_t)(26 + d) * width + k]; D1h[c][d] = snap_in[(int64_t)(28 + d) * width + k]; }
      I2h[c] = snap_in[(int64_t)30 * width + k]; D2h[c] = snap_in[(int64_t)31 * width + k];
      mcur[c] = Mh[c][0][0];
    }
  }
  int s = s0;
  bool done = false;
  const int s_stop = min(s0 + T, J.smax);
  for (int tb = s0; !done && tb < s_stop; tb += NCL) {
#pragma unroll
  for (int jj = 1; jj <= NCL; ++jj) {
    const int cl = jj % NCL;
    const int sn = tb + jj;
    int lM10, lM25, lI1, lI2, rM10, rM25, rD1, rD2;
    const int par = sn & 1;
    if (lane == 63) { int* e = s_edge[par][wv][0]; e[0] = Mh[C - 1][cl][1]; e[1] = Mh[C - 1][cl][4]; e[2] = I1h[C - 1][E1 - 1]; e[3] = I2h[C - 1]; }
    if (lane == 0)  { int* e = s_edge[par][wv][1]; e[0] = Mh[0][cl][1];     e[1] = Mh[0][cl][4];     e[2] = D1h[0][E1 - 1];     e[3] = D2h[0]; }
    lds_barrier();
    done = J.endsfree ? (s_endk != INT32_MAX) : (s_done != 0);
    if (done || sn > s_stop) break;
    s = sn;
    const int lo = max(lo0 - s, max(-pl, kmin)), hi = min(hi0 + s, min(tl, kmax));
    const bool valid = lo <= hi;
    if (!valid || kw_hi < lo - 1 || kw_lo > hi + 1) continue;
    lM10 = from_prev_lane(Mh[C - 1][cl][1]); lM25 = from_prev_lane(Mh[C - 1][cl][4]);
    lI1 = from_prev_lane(I1h[C - 1][E1 - 1]); lI2 = from_prev_lane(I2h[C - 1]);
    rM10 = from_next_lane(Mh[0][cl][1]); rM25 = from_next_lane(Mh[0][cl][4]);
    rD1 = from_next_lane(D1h[0][E1 - 1]); rD2 = from_next_lane(D2h[0]);
    if (lane == 0 && wv > 0) { const int* e = s_edge[par][wv - 1][0]; lM10 = e[0]; lM25 = e[1]; lI1 = e[2]; lI2 = e[3]; }
    if (lane == 63 && wv + 1 < NW) { const int* e = s_edge[par][wv + 1][1]; rM10 = e[0]; rM25 = e[1]; rD1 = e[2]; rD2 = e[3]; }
    int nM[C], nI1[C], nI2[C], nD1[C], nD2[C], preM[C];
    unsigned btb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      const int a10 = c == 0 ? lM10 : Mh[c - 1][cl][1], b10 = c == C - 1 ? rM10 : Mh[c + 1][cl][1];
      const int a25 = c == 0 ? lM25 : Mh[c - 1][cl][4], b25 = c == C - 1 ? rM25 : Mh[c + 1][cl][4];
      const int i1 = c == 0 ? lI1 : I1h[c - 1][E1 - 1], d1 = c == C - 1 ? rD1 : D1h[c + 1][E1 - 1];
      const int i2 = c == 0 ? lI2 : I2h[c - 1], d2 = c == C - 1 ? rD2 : D2h[c + 1];
      const int mx = Mh[c][cl][0];
      unsigned bits = 0;
      if (i1 >= a10) bits |= BT_I1_EXT;
      if (i2 >= a25) bits |= BT_I2_EXT;
      if (d1 >= b10) bits |= BT_D1_EXT;
      if (d2 >= b25) bits |= BT_D2_EXT;
      const unsigned hm = hmaxu[c];
      int ins1 = max(a10, i1) + 1, ins2 = max(a25, i2) + 1, del1 = max(b10, d1), del2 = max(b25, d2), mis = mx + 1;
      ins1 = (unsigned)ins1 <= hm ? ins1 : WF_NULL;
      ins2 = (unsigned)ins2 <= hm ? ins2 : WF_NULL;
      del1 = (unsigned)del1 <= hm ? del1 : WF_NULL;
      del2 = (unsigned)del2 <= hm ? del2 : WF_NULL;
      mis = (unsigned)mis <= hm ? mis : WF_NULL;
      int m = ins1; unsigned src = C_I1;
      if (ins2 >= m) { m = ins2; src = C_I2; }
      if (del1 >= m) { m = del1; src = C_D1; }
      if (del2 >= m) { m = del2; src = C_D2; }
      if (mis >= m)  { m = mis;  src = C_M; }
      const bool on = k >= lo && k <= hi;
      nI1[c] = on ? ins1 : WF_NULL; nI2[c] = on ? ins2 : WF_NULL; nD1[c] = on ? del1 : WF_NULL; nD2[c] = on ? del2 : WF_NULL;
      nM[c] = (on && colok[c]) ? m : WF_NULL;
      preM[c] = nM[c];
      btb[c] = bits | src;
    }
    int ext[C], maxn[C];
    unsigned oP[C], oT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int m = nM[c];
      oP[c] = (unsigned)(m + cP[c]); oT[c] = (unsigned)(m + dT);
      maxn[c] = (int)hmaxu[c] - m;  // (a cell that holds nothing: whatever -- pk_extend2_fast looks at m first)
    }
    pk_extend2_fast(SRC, nM, oP, oT, maxn, wlim, ext, false);
    int32_t* pre = pre_base + (int64_t)s * width;
    uint8_t* bt = bt_base + (int64_t)s * width;
    // (wfa_base2_kernel's step: the lean extension, the end tests only where a cell has reached the far edge of its diagonal)
    bool reached = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int k = k0 + c;
      const bool on = k >= lo && k <= hi;
      if (on) {
        int m = nM[c];
        if (m >= 0) m += ext[c];
        nM[c] = m;
        if (incore[c]) {
          pre[k] = preM[c];
          bt[k] = (uint8_t)btb[c];
          reached |= !J.endsfree || m >= (int)hmaxu[c];
        }
      }
    }
    if (__builtin_expect(__any(reached), 0)) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int k = k0 + c;
        if (k >= lo && k <= hi && incore[c]) end_checks(k, nM[c], nI1[c], nI2[c], nD1[c], nD2[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      mcur[c] = nM[c];
#pragma unroll
      for (int e = DEP - 1; e > 0; --e) Mh[c][cl][e] = Mh[c][cl][e - 1];
      Mh[c][cl][0] = nM[c];
#pragma unroll
      for (int d = E1 - 1; d > 0; --d) { I1h[c][d] = I1h[c][d - 1]; D1h[c][d] = D1h[c][d - 1]; }
      I1h[c][0] = nI1[c]; D1h[c][0] = nD1[c];
      I2h[c] = nI2[c]; D2h[c] = nD2[c];
    }
  }
  }
  __syncthreads();  // (the end tests of the block's last step)
  const bool found = J.endsfree ? (s_endk != INT32_MAX) : (s_done != 0);
  if (found) {
    if (J.endsfree) {
#pragma unroll
      for (int c = 0; c < C; ++c) if (k0 + c == s_endk) s_endoff = mcur[c];
      __syncthreads();
    }
    if (tid == 0) {
      const int ek = J.endsfree ? s_endk : k_end;
      tile_key[blockIdx.x] = ((unsigned long long)(unsigned)s << 32) | (unsigned)(ek - kmin);
      tile_off[blockIdx.x] = J.endsfree ? s_endoff : tl;
    }
    return;
  }
  if (tid == 0) tile_key[blockIdx.x] = ~0ull;
  if (s != s0 + T) return;  // (the budget ended inside the block: nothing follows)
  // ---- the core's columns of the snapshot
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int k = k0 + c;
    if (!incore[c]) continue;
#pragma unroll
    for (int d = 0; d < H; ++d) snap_out[(int64_t)d * width + k] = Mh[c][(NCL - d % NCL) % NCL][d / NCL];
#pragma unroll
    for (int d = 0; d < E1; ++d) { snap_out[(int64_t)(26 + d) * width + k] = I1h[c][d]; snap_out[(int64_t)(28 + d) * width + k] = D1h[c][d]; }
    snap_out[(int64_t)30 * width + k] = I2h[c]; snap_out[(int64_t)31 * width + k] = D2h[c];
  }
}

// Between two blocks: the end test over the tiles of a job -- the first score at which a cell of the row ends the alignment and the smallest such
// diagonal, as the one workgroup of wfa_base2_kernel finds it -- or the next block's score and snapshots.  One thread per job.
__global__ __launch_bounds__(64) void wfa_base2t_advance_kernel(Base2TJob* __restrict__ jobs, const unsigned long long* __restrict__ tile_key,
                                                               const int32_t* __restrict__ tile_off, int njobs, int T, int32_t* __restrict__ active_slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= njobs) return;
  Base2TJob J = jobs[i];
  if (J.done) return;
  unsigned long long best = ~0ull; int at = -1;
  for (int t = 0; t < J.ntiles; ++t) { const unsigned long long k = tile_key[J.task0 + t]; if (k < best) { best = k; at = t; } }
  if (at >= 0) {
    J.done = 1; J.end_s = (int)(best >> 32); J.end_k = (int)(unsigned)(best & 0xffffffffull) + J.b.kmin; J.end_off = tile_off[J.task0 + at];
  } else if (J.s0 + T >= J.b.smax) {
    J.done = 2; J.end_s = J.b.smax;
  } else {
    J.s0 += T;
    const int64_t t = J.snap_in; J.snap_in = J.snap_out; J.snap_out = t;
    atomicAdd(active_slot, 1);
  }
  jobs[i] = J;
}

// The walk back and the result of a job: one wave per job.
__global__ __launch_bounds__(64) void wfa_base2t_finish_kernel(const int32_t* __restrict__ arena32, const uint8_t* __restrict__ arena8, uint32_t* __restrict__ rle,
                                                              const Base2TJob* __restrict__ jobs, BaseResult* __restrict__ results) {
  const Base2TJob JT = jobs[blockIdx.x];
  const BaseJob& J = JT.b;
  const int lane = threadIdx.x;
  const int kmin = J.kmin, kmax = J.kmin + J.width - 1;
  BaseResult r; r.status = JT.done == 1 ? 0 : WFM_DEV_OVERFLOW; r.score = JT.end_s; r.nruns = 0; r.pad_ = 0;
  // the cells of the rows 0 .. end_s as wfa_base2_kernel counts them
  int lo0, hi0;
  if (J.endsfree) { lo0 = max(-J.pbf, kmin); hi0 = min(J.tbf, kmax); }
  else { lo0 = 0; hi0 = 0; }
  unsigned long long cells = 0;
  for (int s = 1 + lane; s <= JT.end_s; s += 64) {
    const int lo = max(lo0 - s, max(-J.pl, kmin)), hi = min(hi0 + s, min(J.tl, kmax));
    if (lo <= hi) cells += (unsigned long long)(hi - lo + 1);
  }
  for (int d = 32; d > 0; d >>= 1) cells += __shfl_down(cells, d, 64);
  cells = __shfl(cells, 0, 64) + (unsigned long long)(hi0 - lo0 + 1);
  r.cells = cells;
  if (JT.done == 1) r.nruns = base2_walk(J, arena32 + J.pre_off - kmin, arena8 + J.bt_off - kmin, rle, JT.end_s, JT.end_k, JT.end_off, lane);
  if (lane == 0) results[blockIdx.x] = r;
}

void launch_base2t_block(const uint32_t* pk, int32_t* a32, uint8_t* a8, const Base2TJob* jobs, const Base2TTask* tasks, unsigned long long* tile_key,
                         int32_t* tile_off, int ntasks, int T, hipStream_t st) {
  hipLaunchKernelGGL(wfa_base2t_kernel<B2T_THREADS>, dim3(ntasks), dim3(B2T_THREADS), 0, st, pk, a32, a8, jobs, tasks, tile_key, tile_off, T);
}
void launch_base2t_advance(Base2TJob* jobs, const unsigned long long* tile_key, const int32_t* tile_off, int njobs, int T, int32_t* active_slot, hipStream_t st) {
  hipLaunchKernelGGL(wfa_base2t_advance_kernel, dim3((njobs + 63) / 64), dim3(64), 0, st, jobs, tile_key, tile_off, njobs, T, active_slot);
}
void launch_base2t_finish(const int32_t* a32, const uint8_t* a8, uint32_t* rle, const Base2TJob* jobs, BaseResult* res, int njobs, hipStream_t st) {
  hipLaunchKernelGGL(wfa_base2t_finish_kernel, dim3(njobs), dim3(64), 0, st, a32, a8, rle, jobs, res);
}

// threads: a multiple of 64 with threads * 2 >= the widest row of the launch (<= 1024)
void launch_base2(const uint32_t* pk, int32_t* a32, uint8_t* a8, uint32_t* rle, const BaseJob* jobs, BaseResult* res, int njobs, int threads, hipStream_t st) {
  if (threads <= 64) hipLaunchKernelGGL(wfa_base2_kernel<64>, dim3(njobs), dim3(64), 0, st, pk, a32, a8, rle, jobs, res);
  else if (threads <= 256) hipLaunchKernelGGL(wfa_base2_kernel<256>, dim3(njobs), dim3(threads), 0, st, pk, a32, a8, rle, jobs, res);
  else hipLaunchKernelGGL(wfa_base2_kernel<1024>, dim3(njobs), dim3(threads), 0, st, pk, a32, a8, rle, jobs, res);
}

// ---------------------------------------------------------------------------
// self-test of the wave shifts the kernel leans on (tests/test_align_gpu.py): out[lane] = value of lane - 1, out[64 + lane] = lane + 1
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dpp_selftest_kernel(int* __restrict__ out) {
  const int lane = threadIdx.x;
  out[lane] = from_prev_lane(1000 + lane);
  out[64 + lane] = from_next_lane(1000 + lane);
}
// ... and of the other property of the target the tile kernels lean on: waves that have ENDED drop out of s_barrier.  256 threads, the waves
// from `keep` on return at once, the others pass values round through LDS over 64 barrier-separated steps (scripts/micro/barrier_exit.hip).
// A target on which ended waves still count would hang here -- under a test's timeout, not inside an alignment.
__global__ __launch_bounds__(256) void barrier_exit_selftest_kernel(int* __restrict__ out, int keep) {
  __shared__ int s[2][4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wv >= keep) return;
  int v = wv + 1;
  for (int t = 0; t < 64; ++t) {
    if (lane == 0) s[t & 1][wv] = v;
    __syncthreads();
    v = (v + s[t & 1][(wv + 1) % keep]) & 0xffff;
  }
  if (lane == 0) out[wv] = v;
}
int selftest_dpp(int* host_out128, hipStream_t st) {
  int* d = nullptr;
  if (hipMalloc((void**)&d, (128 + 16) * sizeof(int)) != hipSuccess) return -1;
  hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, st, d);
  hipError_t e = hipMemcpyAsync(host_out128, d, 128 * sizeof(int), hipMemcpyDeviceToHost, st);
  int got[16] = {0};
  for (int keep = 1; keep <= 3 && e == hipSuccess; ++keep) {
    hipLaunchKernelGGL(barrier_exit_selftest_kernel, dim3(1), dim3(256), 0, st, d + 128 + 4 * (keep - 1), keep);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(got, d + 128, 12 * sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e != hipSuccess) return -1;
  for (int keep = 1; keep <= 3; ++keep) {  // the same exchange on the host
    int v[4] = {1, 2, 3, 4};
    for (int t = 0; t < 64; ++t) {
      int nv[4];
      for (int w = 0; w < keep; ++w) nv[w] = (v[w] + v[(w + 1) % keep]) & 0xffff;
      for (int w = 0; w < keep; ++w) v[w] = nv[w];
    }
    for (int w = 0; w < keep; ++w) if (got[4 * (keep - 1) + w] != v[w]) return -2;
  }
  return 0;
}

}  // namespace wfm
