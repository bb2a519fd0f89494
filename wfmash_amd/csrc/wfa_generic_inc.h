// wfa_generic_inc.h -- the kernels that work for ANY penalties: the step-by-step BiWFA breakpoint kernel, the base kernel
// (unidirectional WFA with backtrace decisions) and the LDS time-tile kernel.  Included twice by wfa_kernels.hip, inside
// namespace wfm::r32 and namespace wfm::r128: RING / RMASK (rows a wavefront ring keeps per component, a power of two >=
// the score scope max(x, o1 + e1, o2 + e2) + 1) resolve to 32 or 128 there.  The default penalties (scope 26) run on
// 32 rows; any -g the reference accepts up to o2 + e2 = 125 runs on 128 (parse_args.hpp:272-288).

struct BpCtx {
  const uint8_t* P[2];
  const uint8_t* T[2];
  int32_t* ring;  // job ring base, already offset so that [row*width + k] works with +pl+1 applied
  int64_t width;
  int pl, tl, koff;
  int kb_lo, kb_hi;  // (tl - pl) -+ the job's score bound
  DevPen pen;
};

__device__ __forceinline__ int32_t* bp_row(const BpCtx& c, int dir, int comp, int s) {
  return c.ring + ((int64_t)((dir * 5 + comp) * RING + (s & RMASK))) * c.width;
}

__device__ __forceinline__ Src bp_src(const BpCtx& c, int dir, int comp, int s, const int (*s_lo)[RING], const int (*s_hi)[RING]) {
  Src r;
  if (s < 0) {
    r.p = c.ring; r.lo = 1; r.hi = 0;
  } else {
    r.p = bp_row(c, dir, comp, s);
    r.lo = s_lo[dir][s & RMASK];
    r.hi = s_hi[dir][s & RMASK];
  }
  return r;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef v4i v4i_u __attribute__((aligned(4)));

// source row for the vector path: p[k] addresses diagonal k; a cell k is live iff
// (unsigned)(k - lo) <= span.  Dead rows: lo = INT32_MIN/2, span = 0.
struct VSrc {
  const int32_t* p;
  int lo;
  unsigned span;
};

__device__ __forceinline__ VSrc bp_vsrc(const BpCtx& c, int dir, int comp, int s, const int (*s_lo)[RING], const int (*s_hi)[RING]) {
  VSrc r;
  r.p = bp_row(c, dir, comp, s);  // always a mapped address of this job's ring
  int lo = 1, hi = 0;
  if (s >= 0) { lo = s_lo[dir][s & RMASK]; hi = s_hi[dir][s & RMASK]; }
  if (lo <= hi) { r.lo = lo; r.span = (unsigned)(hi - lo); }
  else { r.lo = INT32_MIN / 2; r.span = 0u; }
  return r;
}

template <bool MASK>
__device__ __forceinline__ v4i ld4(const VSrc& r, int kb) {
  v4i v = *reinterpret_cast<const v4i_u*>(r.p + kb);
  if (MASK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((unsigned)(kb + j - r.lo) <= r.span) ? v[j] : WF_NULL;
  }
  return v;
}

// 4 consecutive diagonals k0..k0+3 (one 16-byte column chunk) per thread.
struct Cells4 {
  v4i ins1, ins2, del1, del2, m;
  int ext[4], maxn[4];
  bool more[4];
};

// recurrences of 4 consecutive diagonals k0..k0+3 and the first 40 bases of their extensions
template <bool MASK>
__device__ __forceinline__ void bp_cells4_compute(const BpCtx& c, const uint8_t* P, const uint8_t* T, int k0, int lo, int hi,
                                                  const VSrc& mx, const VSrc& mo1, const VSrc& mo2, const VSrc& i1, const VSrc& d1,
                                                  const VSrc& i2, const VSrc& d2, Cells4& q) {
  const unsigned upl = (unsigned)c.pl, utl = (unsigned)c.tl;
  const v4i a1 = ld4<MASK>(mo1, k0 - 1), b1 = ld4<MASK>(mo1, k0 + 1);
  const v4i a2 = ld4<MASK>(mo2, k0 - 1), b2 = ld4<MASK>(mo2, k0 + 1);
  const v4i vi1 = ld4<MASK>(i1, k0 - 1), vd1 = ld4<MASK>(d1, k0 + 1);
  const v4i vi2 = ld4<MASK>(i2, k0 - 1), vd2 = ld4<MASK>(d2, k0 + 1);
  const v4i vmx = ld4<MASK>(mx, k0);
  uint64_t x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    q.ins1[j] = valid_or_null(max(a1[j], vi1[j]) + 1, k, upl, utl);
    q.ins2[j] = valid_or_null(max(a2[j], vi2[j]) + 1, k, upl, utl);
    q.del1[j] = valid_or_null(max(b1[j], vd1[j]), k, upl, utl);
    q.del2[j] = valid_or_null(max(b2[j], vd2[j]), k, upl, utl);
    const int mis = valid_or_null(vmx[j] + 1, k, upl, utl);
    int mm = max(imax3(q.ins1[j], q.ins2[j], mis), max(q.del1[j], q.del2[j]));
    if (k < lo || k > hi) mm = WF_NULL;  // edge chunk: cells outside the row are dead
    q.m[j] = mm;
    // first 8 bases of the extension for all four cells (independent loads)
    x[j] = 0; q.maxn[j] = 0;
    if (mm >= 0) {
      q.maxn[j] = min(c.pl - (mm - k), c.tl - mm);
      x[j] = load8(P + (mm - k)) ^ load8(T + mm);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    q.ext[j] = 0; q.more[j] = false;
    if (q.m[j] >= 0) q.ext[j] = lce_head40_g(P, T, q.m[j] - (k0 + j), q.m[j], x[j], q.maxn[j], q.more[j]);
  }
}

// Computes + extends row s of direction dir.  Returns #cells of the row (uniform).
// Per-thread max antidiagonal is accumulated into mak.
template <bool TRACK = false>
__device__ __forceinline__ int bp_compute_row(const BpCtx& c, int dir, int s, int (*s_lo)[RING], int (*s_hi)[RING], int& mak, long long* sec,
                                              int* cmax = nullptr) {
  const DevPen& pn = c.pen;
  const VSrc mx  = bp_vsrc(c, dir, C_M,  s - pn.x, s_lo, s_hi);
  const VSrc mo1 = bp_vsrc(c, dir, C_M,  s - pn.o1 - pn.e1, s_lo, s_hi);
  const VSrc mo2 = bp_vsrc(c, dir, C_M,  s - pn.o2 - pn.e2, s_lo, s_hi);
  const VSrc i1  = bp_vsrc(c, dir, C_I1, s - pn.e1, s_lo, s_hi);
  const VSrc d1  = bp_vsrc(c, dir, C_D1, s - pn.e1, s_lo, s_hi);
  const VSrc i2  = bp_vsrc(c, dir, C_I2, s - pn.e2, s_lo, s_hi);
  const VSrc d2  = bp_vsrc(c, dir, C_D2, s - pn.e2, s_lo, s_hi);
  int lo = INT32_MAX, hi = INT32_MIN;
  // interior = diagonals whose k-1..k+1 neighbourhood is live in EVERY source row
  int in_lo = INT32_MIN, in_hi = INT32_MAX;
  bool all_live = true;
#define ROW_RANGE(r, dl, dh)                                                               \
  if ((r).span != 0u || (r).lo != INT32_MIN / 2) {                                         \
    lo = min(lo, (r).lo + (dl)); hi = max(hi, (r).lo + (int)(r).span + (dh));              \
    in_lo = max(in_lo, (r).lo); in_hi = min(in_hi, (r).lo + (int)(r).span);                \
  } else all_live = false;
  ROW_RANGE(mx, 0, 0) ROW_RANGE(mo1, -1, 1) ROW_RANGE(mo2, -1, 1) ROW_RANGE(i1, -1, 1) ROW_RANGE(i2, -1, 1)
#undef ROW_RANGE
  if (d1.span == 0u && d1.lo == INT32_MIN / 2) all_live = false;
  lo = max(max(lo, -c.pl), c.kb_lo + s);  // and only what can still reach the end diagonal within the score bound (see Rng)
  hi = min(min(hi, c.tl), c.kb_hi - s);
  const bool valid = (lo <= hi);
  if (threadIdx.x == 0) {
    s_lo[dir][s & RMASK] = valid ? lo : 1;
    s_hi[dir][s & RMASK] = valid ? hi : 0;
  }
  if (!valid) return 0;
  int32_t* om  = bp_row(c, dir, C_M, s);
  int32_t* oi1 = bp_row(c, dir, C_I1, s);
  int32_t* oi2 = bp_row(c, dir, C_I2, s);
  int32_t* od1 = bp_row(c, dir, C_D1, s);
  int32_t* od2 = bp_row(c, dir, C_D2, s);
  const uint8_t* P = c.P[dir];
  const uint8_t* T = c.T[dir];
  const int koff = c.koff;  // column = k + koff, multiple-of-4 columns are 16-byte aligned
  const int c_lo = (lo + koff) >> 2, c_hi = (hi + koff) >> 2;
  // the chunk loop is uniform over the workgroup (lanes past the row's end idle through it): the long extensions are
  // finished by whole waves (wave_lce_tail_g)
  for (int chb = c_lo; chb <= c_hi; chb += (int)blockDim.x) {
    const int ch = chb + (int)threadIdx.x;
    const bool on = ch <= c_hi;
    const int k0 = (ch << 2) - koff;
    Cells4 q;
#pragma unroll
    for (int j = 0; j < 4; ++j) { q.m[j] = WF_NULL; q.ext[j] = 0; q.maxn[j] = 0; q.more[j] = false; }
    if (on) {
      // loads touch k0-1 .. k0+4
      if (all_live && k0 - 1 >= in_lo && k0 + 4 <= in_hi) bp_cells4_compute<false>(c, P, T, k0, lo, hi, mx, mo1, mo2, i1, d1, i2, d2, q);
      else bp_cells4_compute<true>(c, P, T, k0, lo, hi, mx, mo1, mo2, i1, d1, i2, d2, q);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (__any(q.more[j])) q.ext[j] = wave_lce_tail_g(P, T, q.m[j] - (k0 + j), q.m[j], q.ext[j], q.maxn[j], q.more[j]);
    if (on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (q.m[j] >= 0) {
          q.m[j] += min(q.ext[j], q.maxn[j]);
          mak = max(mak, 2 * q.m[j] - (k0 + j));
        }
      }
      if (TRACK) {  // per-component row maxima for the phase-2 overlap pruning, while the cells are in registers
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k0 + j >= lo && k0 + j <= hi) {
            cmax[C_M] = max(cmax[C_M], q.m[j]); cmax[C_I1] = max(cmax[C_I1], q.ins1[j]); cmax[C_I2] = max(cmax[C_I2], q.ins2[j]);
            cmax[C_D1] = max(cmax[C_D1], q.del1[j]); cmax[C_D2] = max(cmax[C_D2], q.del2[j]);
          }
        }
      }
      *reinterpret_cast<v4i*>(oi1 + k0) = q.ins1;
      *reinterpret_cast<v4i*>(oi2 + k0) = q.ins2;
      *reinterpret_cast<v4i*>(od1 + k0) = q.del1;
      *reinterpret_cast<v4i*>(od2 + k0) = q.del2;
      *reinterpret_cast<v4i*>(om + k0) = q.m;
    }
  }
  return hi - lo + 1;
}

// wavefront_bialign_overlap, data-parallel part: for every (i, comp) whose
// breakpoint score would beat `best`, find the smallest k0 with
// off0[k0] + off1[k1] >= tl.  Results in s_mink[i*5+comp].
// Per-row, per-component maximum offset (0 if the row holds no live cell): an upper
// bound that lets the overlap test skip every (score, component) pair whose rows cannot
// reach off0 + off1 >= tl anywhere.  Pure pruning: results are unchanged.
__device__ __forceinline__ void bp_row_maxima(const BpCtx& c, int d, int s, const int (*s_lo)[RING], const int (*s_hi)[RING],
                                              int (*s_rmax)[RING][5]) {
  const int lo = s_lo[d][s & RMASK], hi = s_hi[d][s & RMASK];
  int mx[5] = {0, 0, 0, 0, 0};
  if (lo <= hi) {
    // 16-byte column chunks, the five components of a chunk in flight together; cells outside
    // [lo, hi] are masked (they may hold stale values of an older row)
    const int koff = c.koff;
    const int c_lo = (lo + koff) >> 2, c_hi = (hi + koff) >> 2;
    const int32_t* r[5];
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) r[cc] = bp_row(c, d, cc, s);
    for (int ch = c_lo + (int)threadIdx.x; ch <= c_hi; ch += (int)blockDim.x) {
      const int k0 = (ch << 2) - koff;
      v4i v[5];
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) v[cc] = *reinterpret_cast<const v4i_u*>(r[cc] + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k0 + j >= lo && k0 + j <= hi) {
#pragma unroll
          for (int cc = 0; cc < 5; ++cc) mx[cc] = max(mx[cc], v[cc][j]);
        }
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) {
    const int v = wave_max_dpp63(mx[cc]);
    if ((threadIdx.x & 63) == 63 && v > 0) atomicMax(&s_rmax[d][s & RMASK][cc], v);
  }
}


// wavefront_bialign_overlap, data-parallel part: for every (i, comp) whose
// breakpoint score would beat `best`, find the smallest k0 with
// off0[k0] + off1[k1] >= tl.  Results in s_mink[i*5+comp].
// which components of row s1 - i of the other direction can still give a better breakpoint with row s0 (bit cc): the score
// test of the reference's loop and the row maxima (a pair whose largest offsets cannot reach tl meets nowhere)
__device__ __forceinline__ unsigned bp_pair_bits(const BpCtx& c, int d0, int s0, int s1, int i, int best, const int (*s_lo)[RING], const int (*s_hi)[RING],
                                                 const int (*s_rmax)[RING][5]) {
  const int d1 = d0 ^ 1, si = s1 - i;
  if (si < 0) return 0u;
  if (s0 + si - c.pen.o2 >= best) return 0u;
  if (s_lo[d1][si & RMASK] > s_hi[d1][si & RMASK]) return 0u;
  unsigned bits = 0;
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) {
    if (s0 + si - bp_gap_open(c.pen, cc) >= best) continue;
    if (s_rmax[d0][s0 & RMASK][cc] + s_rmax[d1][si & RMASK][cc] < c.tl) continue;
    bits |= 1u << cc;
  }
  return bits;
}

__device__ __forceinline__ void bp_overlap_scan(const BpCtx& c, int d0, int s0, int s1, int best,
                                                const int (*s_lo)[RING], const int (*s_hi)[RING], int* s_mink, int scope,
                                                const int (*s_rmax)[RING][5], const unsigned char* s_bits) {
  const int d1 = d0 ^ 1;
  const int lo0 = s_lo[d0][s0 & RMASK], hi0 = s_hi[d0][s0 & RMASK];
  if (lo0 > hi0) return;
  const int kinv = c.tl - c.pl;
  // s_bits[i] (bp_pair_bits, filled by the caller before its barrier): the candidate pairs of this test, uniform
  int klo = INT32_MAX, khi = INT32_MIN;
  int rm1[5] = {0, 0, 0, 0, 0};  // largest opposite-direction offset any active row holds, per component
  bool any = false;
  for (int i = 0; i < scope; ++i) {
    const unsigned bits = s_bits[i];
    if (!bits) continue;
    any = true;
    const int si = s1 - i;
#pragma unroll
    for (int cc = 0; cc < 5; ++cc)
      if (bits & (1u << cc)) rm1[cc] = max(rm1[cc], s_rmax[d1][si & RMASK][cc]);
    klo = min(klo, kinv - s_hi[d1][si & RMASK]); khi = max(khi, kinv - s_lo[d1][si & RMASK]);
  }
  if (!any) return;
  klo = max(klo, lo0); khi = min(khi, hi0);
  const int32_t* r0[5];
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) r0[cc] = bp_row(c, d0, cc, s0);
  // 16-byte chunks of the new row, five components in flight together
  const int c_lo = (klo + c.koff) >> 2, c_hi = (khi + c.koff) >> 2;
  for (int ch = c_lo + (int)threadIdx.x; ch <= c_hi; ch += (int)blockDim.x) {
    const int kb = (ch << 2) - c.koff;
    v4i v[5];
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) v[cc] = *reinterpret_cast<const v4i_u*>(r0[cc] + kb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = kb + j;
      if (k0 < klo || k0 > khi) continue;
      const int k1 = kinv - k0;
      int o0[5];
      bool reach = false;  // can this diagonal meet ANY active opposite row?  (most diagonals cannot: pure pruning)
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) {
        o0[cc] = v[cc][j];
        reach = reach || (o0[cc] >= 0 && o0[cc] + rm1[cc] >= c.tl);
      }
      if (!reach) continue;
      for (int i = 0; i < scope; ++i) {
        const unsigned bits = s_bits[i];
        if (!bits) continue;
        const int si = s1 - i;
        const int lo1 = s_lo[d1][si & RMASK], hi1 = s_hi[d1][si & RMASK];
        if (k1 < lo1 || k1 > hi1) continue;
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) {
          if (!(bits & (1u << cc))) continue;
          if (o0[cc] < 0 || o0[cc] + s_rmax[d1][si & RMASK][cc] < c.tl) continue;
          const int o1 = bp_row(c, d1, cc, si)[k1];
          if (o0[cc] + o1 >= c.tl) atomicMin(&s_mink[i * 5 + cc], k0);
        }
      }
    }
  }
}

// Row 0 of one direction (wavefront_unialign_init, end2end): the begin component holds
// offset 0 at k = 0, M is extended.  Returns 1 if the alignment already ends at score 0.
// Called by one whole wave per direction (the first extension of a sub-problem is often its longest: the wave walks
// it together); the results are valid in every lane.
__device__ __forceinline__ int bp_init_row0(const BpCtx& c, int d, int cb, int ce, int& mak) {
  int m0 = WF_NULL;
  mak = 0;
  const bool first = (threadIdx.x & 63u) == 0;
  if (first) for (int cc = 1; cc < 5; ++cc) bp_row(c, d, cc, 0)[0] = (cc == cb) ? 0 : WF_NULL;
  if (cb == C_M) {
    m0 = wave_lce(c.P[d], c.T[d], 0, 0, min(c.pl, c.tl), first);
    m0 = rdlane(m0, 0);
    mak = 2 * m0;
  }
  if (first) bp_row(c, d, C_M, 0)[0] = m0;
  // termination at score 0 (identical sequences): only possible for M/M forms
  return (c.pl == c.tl && ce == C_M && m0 >= c.tl) ? 1 : 0;
}

__global__ __launch_bounds__(1024) void wfa_bp_kernel(const uint8_t* __restrict__ seq, int32_t* __restrict__ ring_arena,
                                                      const BpJob* __restrict__ jobs, BpResult* __restrict__ results,
                                                      DevPen pen, int scope) {
  const BpJob J = jobs[blockIdx.x];
  __shared__ int s_lo[2][RING];
  __shared__ int s_hi[2][RING];
  __shared__ int s_mak[3][2];
  __shared__ int s_mink[RING * 5];
  __shared__ unsigned char s_bits[RING];  // per row of the other direction: the components that can still matter in the test at hand
  __shared__ int s_bp[8];  // score, score_fwd, score_rev, k_fwd, off_fwd, comp
  __shared__ int s_rmax[2][RING][5];

  if (J.resume_s == -3) {  // the tile phase ran out of the job's band: nothing to do here, the host retries on a full ring
    if (threadIdx.x == 0) {
      BpResult r; r.status = WFM_DEV_BAND; r.score = 0; r.score_fwd = 0; r.score_rev = 0; r.k_fwd = 0; r.off_fwd = 0; r.comp = 0; r.steps = 0; r.cells = 0;
      r.steps_p1 = 0; r.ticks_p1 = 0; r.ticks_p2 = 0; r.pad_ = 0;
      results[blockIdx.x] = r;
    }
    return;
  }
  const int band = J.band;
  BpCtx c;
  c.P[0] = seq + J.p_fwd; c.T[0] = seq + J.t_fwd;
  c.P[1] = seq + J.p_rev; c.T[1] = seq + J.t_rev;
  c.width = J.width;
  c.koff = J.koff;  // columns start at 4: chunk 0 is never touched, so k0-1 loads stay inside the row
  c.ring = ring_arena + J.ring_off + c.koff;
  c.pl = J.pl; c.tl = J.tl;
  c.kb_lo = (J.tl - J.pl) - J.sub; c.kb_hi = (J.tl - J.pl) + J.sub;
  c.pen = pen;
  const int tid = threadIdx.x;
  const int A = J.pl + J.tl - 1;  // max_antidiagonal
  uint64_t cells = 0;

  // ---- init rows 0 (wavefront_unialign_init, end2end), or resume from a tiled snapshot ----
  if (tid < 2 * RING) { s_lo[tid / RING][tid % RING] = 1; s_hi[tid / RING][tid % RING] = 0; }
  if (tid < 6) ((int*)s_mak)[tid] = 0;
  for (int i = tid; i < 2 * RING * 5; i += blockDim.x) ((int*)s_rmax)[i] = 0;
  __syncthreads();
  int end_reached = 0;
  const bool exact = J.resume_s >= 0 && J.resume_sr >= 0;  // the snapshot IS the meeting point: phase 1 is over
  const int rs_f = J.resume_s, rs_r = exact ? J.resume_sr : J.resume_s;
  if (J.resume_s >= 0) {
    if (tid < 2 * RING) {
      const int d = tid / RING, sc = (d == 0 ? rs_f : rs_r) - (tid % RING);
      if (sc >= 0) { const Rng RG = make_rng(J.pl, J.tl, J.sub); s_lo[d][sc & RMASK] = rng_lo(RG, sc); s_hi[d][sc & RMASK] = rng_hi(RG, sc); }
    }
    if (tid == 0) { s_mak[0][0] = J.fmax0; s_mak[0][1] = J.rmax0; s_bp[6] = 0; s_bp[7] = 0; }
  } else if (tid < 128) {  // wave 0: forward, wave 1: reverse
    const int d = tid >> 6;
    int mak = 0;
    const int ended = bp_init_row0(c, d, d == 0 ? J.comp_begin : J.comp_end, d == 0 ? J.comp_end : J.comp_begin, mak);
    if ((tid & 63) == 0) {
      s_bp[6 + d] = ended;
      s_lo[d][0] = 0; s_hi[d][0] = 0;
      s_mak[0][d] = mak;
    }
  }
  __syncthreads();
  int fmax = s_mak[0][0], rmax = s_mak[0][1];
  end_reached = s_bp[6] | s_bp[7];
  if (end_reached) {
    if (tid == 0) {
      BpResult r; r.status = 1; r.score = 0; r.score_fwd = 0; r.score_rev = 0; r.k_fwd = 0; r.off_fwd = 0; r.comp = 0; r.steps = 0; r.cells = 2;
      r.steps_p1 = 0; r.ticks_p1 = 0; r.ticks_p2 = 0; r.pad_ = 0;
      results[blockIdx.x] = r;
    }
    return;
  }
  cells = J.resume_s >= 0 ? 0 : 2;
  const int64_t max_steps = (int64_t)(pen.o1 + pen.o2) * 4 + (int64_t)(J.pl + J.tl + 2) * max(pen.x, max(pen.e1, pen.e2)) * 2 + 256;
  int sf = max(rs_f, 0), sr = max(rs_r, 0);
  int last_fwd = exact ? J.last_fwd : 0;
  int status = 0;
  int buf = 0;
  const long long t_begin = wall_clock64();
  long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)sec;

  // ---- phase 1: advance both directions until the antidiagonals meet ----
  // (the reference alternates forward, reverse; both rows of a round are
  //  computed here in one pass and the checks replayed in the same order)
  for (;;) {
    if (fmax + rmax >= A) break;
    if (band > 0 && max(sf, sr) + 2 > band) { status = WFM_DEV_BAND; break; }
    buf = (buf + 1) % 3;
    if (tid == 0) { s_mak[(buf + 1) % 3][0] = 0; s_mak[(buf + 1) % 3][1] = 0; }
    // per-component row maxima (phase-2 pruning) ride along: the slots of the rows computed NEXT
    // round are cleared now, one barrier ahead of the atomics that fill them
    if (tid < 10) s_rmax[tid / 5][((tid < 5 ? sf : sr) + 2) & RMASK][tid % 5] = 0;
    int makf = 0, makr = 0;
    int cmf[5] = {0, 0, 0, 0, 0}, cmr[5] = {0, 0, 0, 0, 0};
    SEC_T(ta);
    const int nf = bp_compute_row<true>(c, 0, sf + 1, s_lo, s_hi, makf, sec, cmf);
    const int nr = bp_compute_row<true>(c, 1, sr + 1, s_lo, s_hi, makr, sec, cmr);
    SEC_T(tb);
    makf = wave_max(makf);
    makr = wave_max(makr);
    if ((tid & 63) == 0) {
      if (makf > 0) atomicMax(&s_mak[buf][0], makf);
      if (makr > 0) atomicMax(&s_mak[buf][1], makr);
    }
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) {
      const int vf = wave_max_dpp63(cmf[cc]), vr = wave_max_dpp63(cmr[cc]);
      if ((tid & 63) == 63) {
        if (vf > 0) atomicMax(&s_rmax[0][(sf + 1) & RMASK][cc], vf);
        if (vr > 0) atomicMax(&s_rmax[1][(sr + 1) & RMASK][cc], vr);
      }
    }
    SEC_T(tc);
    __syncthreads();
    SEC_T(td);
    SEC_ADD(3, ta, tb); SEC_ADD(4, tb, tc); SEC_ADD(5, tc, td);
    ++sf; cells += (uint64_t)nf;
    fmax = max(fmax, s_mak[buf][0]);
    last_fwd = 1;
    if (fmax + rmax >= A) break;  // reverse row sr+1 stays speculative; recomputed in phase 2
    ++sr; cells += (uint64_t)nr;
    rmax = max(rmax, s_mak[buf][1]);
    last_fwd = 0;
    if ((int64_t)sf + sr > max_steps) { status = WFM_DEV_UNREACHABLE; break; }
  }

  // ---- phase 2: overlap detection (wavefront_bialign_find_breakpoint, 2nd loop) ----
  // With a bound of the job's score the loop starts as if a breakpoint of score bound + 1 were in hand: pairs that cannot do
  // better are never looked at, and the loop ends where none can follow.  The bound of a child is exact, so its breakpoint
  // is below it; a root whose guess was too small ends here without one and is run again (WFM_DEV_BAND).
  const int best0 = min(J.sub < SUB_NONE ? J.sub + 1 : INT32_MAX, J.best0 > 0 ? J.best0 : INT32_MAX);  // (or the breakpoint earlier rounds of phase 2 found)
  int best = best0;
  if (tid == 0) s_bp[7] = 0;  // a real breakpoint has been taken
  const long long t_mid = wall_clock64();
#ifdef WFM_PROFILE_SECTIONS
  if (tid == 0 && blockIdx.x == 0) { for (int q = 0; q < 6; ++q) g_sec[q] = sec[q]; g_sec[6] = sf + sr; }
#endif
  const int steps_p1 = sf + sr;
  if (status == 0) {
    const int gopen = max(pen.o1, pen.o2);
    // row maxima of the `scope` newest rows of both directions: rows computed by this kernel
    // already have theirs; rows taken over from a tile snapshot (or row 0) are scanned here
    const int own_f = max(rs_f, 0) + 1, own_r = max(rs_r, 0) + 1;
    for (int i = 0; i < scope; ++i) {
      if (sf - i >= 0 && sf - i < own_f) bp_row_maxima(c, 0, sf - i, s_lo, s_hi, s_rmax);
      if (sr - i >= 0 && sr - i < own_r) bp_row_maxima(c, 1, sr - i, s_lo, s_hi, s_rmax);
    }
    __syncthreads();
    for (;;) {
      int d0;  // direction whose newest wavefront is tested, then the OTHER one advances
      if (last_fwd) {
        const int min_sr = (sr > scope - 1) ? sr - (scope - 1) : 0;
        if (sf + min_sr - gopen >= best) break;
        d0 = 0;
      } else {
        const int min_sf = (sf > scope - 1) ? sf - (scope - 1) : 0;
        if (min_sf + sr - gopen >= best) break;
        d0 = 1;
      }
      const int s0 = d0 == 0 ? sf : sr;
      const int s1 = d0 == 0 ? sr : sf;
      // overlap(a_d0 new row s0, other direction rows s1..s1-scope+1)
      const bool m0_valid = s_lo[d0][s0 & RMASK] <= s_hi[d0][s0 & RMASK];
      if (m0_valid) {
        for (int i = tid; i < RING * 5; i += blockDim.x) s_mink[i] = INT32_MAX;
        if (tid < scope) s_bits[tid] = (unsigned char)bp_pair_bits(c, d0, s0, s1, tid, best, s_lo, s_hi, s_rmax);
        __syncthreads();
        bp_overlap_scan(c, d0, s0, s1, best, s_lo, s_hi, s_mink, scope, s_rmax, s_bits);
        __syncthreads();
        if (tid == 0) {
          int b = best;
          const int order[5] = {C_D2, C_I2, C_D1, C_I1, C_M};
          for (int i = 0; i < scope; ++i) {
            const int si = s1 - i;
            if (si < 0) break;
            for (int oi = 0; oi < 5; ++oi) {
              const int cc = order[oi];
              const int gop = (cc == C_M) ? 0 : ((cc == C_I1 || cc == C_D1) ? pen.o1 : pen.o2);
              // nested `continue`s of wavefront_bialign_overlap: a failed test skips the rest of this i
              if ((oi == 0 || oi == 2 || oi == 4) && s0 + si - gop >= b) break;
              const int k0 = s_mink[i * 5 + cc];
              if (k0 == INT32_MAX) continue;
              if (s0 + si - gop >= b) continue;
              const int k1 = (c.tl - c.pl) - k0;
              const int o0 = bp_row(c, d0, cc, s0)[k0];
              const int o1 = bp_row(c, d0 ^ 1, cc, si)[k1];
              b = s0 + si - gop;
              s_bp[0] = b;
              if (d0 == 0) { s_bp[1] = s0; s_bp[2] = si; s_bp[3] = k0; s_bp[4] = o0; }
              else         { s_bp[1] = si; s_bp[2] = s0; s_bp[3] = k1; s_bp[4] = o1; }
              s_bp[5] = cc;
              s_bp[7] = 1;
            }
          }
          s_bp[6] = b;
        }
        __syncthreads();
        best = s_bp[6];
      }
      // advance the other direction
      if (band > 0 && max(sf, sr) + 2 > band) { status = WFM_DEV_BAND; break; }
      int mak = 0;
      if (tid < 5) s_rmax[d0 ^ 1][((d0 == 0 ? sr : sf) + 1) & RMASK][tid] = 0;
      __syncthreads();  // the clear must not race with the atomics of faster waves below
      int cmax[5] = {0, 0, 0, 0, 0};
      if (d0 == 0) { ++sr; cells += (uint64_t)bp_compute_row<true>(c, 1, sr, s_lo, s_hi, mak, sec, cmax); last_fwd = 0; }
      else         { ++sf; cells += (uint64_t)bp_compute_row<true>(c, 0, sf, s_lo, s_hi, mak, sec, cmax); last_fwd = 1; }
      {  // row maxima of the new row straight from the registers that computed it
        const int sn = d0 == 0 ? sr : sf;
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) {
          const int v = wave_max_dpp63(cmax[cc]);
          if ((tid & 63) == 63 && v > 0) atomicMax(&s_rmax[d0 ^ 1][sn & RMASK][cc], v);
        }
      }
      __syncthreads();
      if (d0 == 1 && (int64_t)sf + sr > max_steps && best == INT32_MAX) { status = WFM_DEV_UNREACHABLE; break; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    BpResult r;
    r.status = status;
    if (status == 0 && best == INT32_MAX) r.status = WFM_DEV_UNREACHABLE;
    if (status == 0 && best0 != INT32_MAX && !s_bp[7]) r.status = J.best0 > 0 ? WFM_DEV_P2_NOTHING : WFM_DEV_BAND;  // nothing within the bound / nothing better
    if (status == WFM_DEV_BAND) r.status = WFM_DEV_BAND;  // a breakpoint found so far may not be the best one
    r.score = best; r.score_fwd = s_bp[1]; r.score_rev = s_bp[2]; r.k_fwd = s_bp[3]; r.off_fwd = s_bp[4]; r.comp = s_bp[5];
    r.steps = sf + sr;
    r.cells = cells;
    r.steps_p1 = steps_p1;
    r.ticks_p1 = (uint32_t)(t_mid - t_begin);
    r.ticks_p2 = (uint32_t)(wall_clock64() - t_mid);
    r.pad_ = 0;
    results[blockIdx.x] = r;
  }
}

// ---------------------------------------------------------------------------
// Base kernel: unidirectional WFA with per-cell backtrace decisions
// ---------------------------------------------------------------------------
struct BaseCtx {
  const uint8_t* P;
  const uint8_t* T;
  int32_t* ring;  // [5][RING][width], indexable by k (kmin applied)
  int32_t* pre;   // [(smax+1)][width] pre-extend M offsets, indexable by k
  uint8_t* bt;    // [(smax+1)][width]
  int64_t width;
  int pl, tl, kmin, kmax;
  DevPen pen;
};

__device__ __forceinline__ int32_t* bs_row(const BaseCtx& c, int comp, int s) {
  return c.ring + ((int64_t)(comp * RING + (s & RMASK))) * c.width;
}
__device__ __forceinline__ Src bs_src(const BaseCtx& c, int comp, int s, const int* s_lo, const int* s_hi) {
  Src r;
  if (s < 0) { r.p = c.ring; r.lo = 1; r.hi = 0; }
  else { r.p = bs_row(c, comp, s); r.lo = s_lo[s & RMASK]; r.hi = s_hi[s & RMASK]; }
  return r;
}

struct RleWriter {
  uint32_t* base;  // entries are written at base[-1], base[-2], ...
  int n;
  int cur_op;
  uint32_t cur_len;
  bool writes = true;  // several lanes may keep the same writer in step; one of them stores
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

// NT: threads of a workgroup.  256 for leaves and ordinary patches (rows of a few hundred to 1.3 k diagonals); 1024 for the
// jobs with wide rows -- a patch eroded to its 4096-base limit starts 8 k diagonals wide, and the ones that overflow
// their first score budget are exactly those: at 256 threads a step walked its row in 30 rounds of dependent loads and a
// handful of such jobs ran 5 - 10 ms behind everybody else's 0.3.
template <int NT>
__global__ __launch_bounds__(NT) void wfa_base_kernel(const uint8_t* __restrict__ seq, int32_t* __restrict__ arena32,
                                                       uint8_t* __restrict__ arena8, uint32_t* __restrict__ rle,
                                                       const BaseJob* __restrict__ jobs, BaseResult* __restrict__ results,
                                                       DevPen pen) {
  const BaseJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  if (J.type != 0) {  // trivial: all-D or all-I (wavefront_bialign_alignment trivial cases)
    if (tid == 0) {
      BaseResult r; r.status = 0; r.cells = 0; r.nruns = 0; r.score = 0;
      const int len = J.type == 1 ? J.pl : J.tl;
      if (len > 0) { rle[J.rle_end - 1] = ((uint32_t)len << 2) | (uint32_t)(J.type == 1 ? OP_D : OP_I); r.nruns = 1; }
      results[blockIdx.x] = r;
    }
    return;
  }
  __shared__ int s_lo[RING];
  __shared__ int s_hi[RING];
  __shared__ int s_done;  // end2end: 1 when reached
  __shared__ int s_endk;  // ends-free: min k satisfying the end condition

  BaseCtx c;
  c.P = seq + J.p_off; c.T = seq + J.t_off;
  c.width = J.width;
  c.ring = arena32 + J.ring_off - J.kmin;
  c.pre = arena32 + J.pre_off - J.kmin;
  c.bt = arena8 + J.bt_off - J.kmin;
  c.pl = J.pl; c.tl = J.tl; c.kmin = J.kmin; c.kmax = J.kmin + J.width - 1;
  c.pen = pen;
  const unsigned upl = (unsigned)c.pl, utl = (unsigned)c.tl;
  const int k_end = c.tl - c.pl;
  uint64_t cells = 0;

  if (tid < RING) { s_lo[tid] = 1; s_hi[tid] = 0; }
  if (tid == 0) { s_done = 0; s_endk = INT32_MAX; }
  __syncthreads();

  // ---- row 0 ----
  int lo0, hi0;
  if (J.endsfree) { lo0 = max(-J.pbf, c.kmin); hi0 = min(J.tbf, c.kmax); }
  else { lo0 = 0; hi0 = 0; }
  for (int kb = lo0; kb <= hi0; kb += blockDim.x) {  // uniform over the workgroup: wave_lce is a wave-wide call
    const int k = kb + tid;
    const bool on = k <= hi0;
    int m = WF_NULL;
    int vi1 = WF_NULL, vi2 = WF_NULL, vd1 = WF_NULL, vd2 = WF_NULL;
    if (J.endsfree) m = k > 0 ? k : 0;
    else {
      if (J.comp_begin == C_M) m = 0;
      vi1 = J.comp_begin == C_I1 ? 0 : WF_NULL;
      vi2 = J.comp_begin == C_I2 ? 0 : WF_NULL;
      vd1 = J.comp_begin == C_D1 ? 0 : WF_NULL;
      vd2 = J.comp_begin == C_D2 ? 0 : WF_NULL;
    }
    if (!on) m = WF_NULL;
    const int ext0 = wave_lce(c.P, c.T, m - k, m, min(c.pl - (m - k), c.tl - m), m >= 0);
    if (!on) continue;
    c.pre[k] = m; c.bt[k] = 0;
    if (m >= 0) {
      m += ext0;
      if (J.endsfree) {
        const int h = m, v = m - k;
        if ((h >= c.tl && c.pl - v <= J.pef) || (v >= c.pl && c.tl - h <= J.tef)) atomicMin(&s_endk, k);
      } else if (k == k_end && J.comp_end == C_M && m >= c.tl) s_done = 1;
    }
    bs_row(c, C_M, 0)[k] = m;
    bs_row(c, C_I1, 0)[k] = vi1; bs_row(c, C_I2, 0)[k] = vi2;
    bs_row(c, C_D1, 0)[k] = vd1; bs_row(c, C_D2, 0)[k] = vd2;
  }
  if (tid == 0) { s_lo[0] = lo0; s_hi[0] = hi0; }
  cells += (uint64_t)(hi0 - lo0 + 1);
  __syncthreads();

  int s = 0;
  int status = 0;
  bool done = J.endsfree ? (s_endk != INT32_MAX) : (s_done != 0);
  while (!done) {
    ++s;
    if (s > J.smax) { status = WFM_DEV_OVERFLOW; break; }
    const DevPen& pn = c.pen;
    const Src mx  = bs_src(c, C_M,  s - pn.x, s_lo, s_hi);
    const Src mo1 = bs_src(c, C_M,  s - pn.o1 - pn.e1, s_lo, s_hi);
    const Src mo2 = bs_src(c, C_M,  s - pn.o2 - pn.e2, s_lo, s_hi);
    const Src i1  = bs_src(c, C_I1, s - pn.e1, s_lo, s_hi);
    const Src d1  = bs_src(c, C_D1, s - pn.e1, s_lo, s_hi);
    const Src i2  = bs_src(c, C_I2, s - pn.e2, s_lo, s_hi);
    const Src d2  = bs_src(c, C_D2, s - pn.e2, s_lo, s_hi);
    int lo = INT32_MAX, hi = INT32_MIN;
    if (mx.lo <= mx.hi)   { lo = min(lo, mx.lo);      hi = max(hi, mx.hi); }
    if (mo1.lo <= mo1.hi) { lo = min(lo, mo1.lo - 1); hi = max(hi, mo1.hi + 1); }
    if (mo2.lo <= mo2.hi) { lo = min(lo, mo2.lo - 1); hi = max(hi, mo2.hi + 1); }
    if (i1.lo <= i1.hi)   { lo = min(lo, i1.lo - 1);  hi = max(hi, i1.hi + 1); }
    if (i2.lo <= i2.hi)   { lo = min(lo, i2.lo - 1);  hi = max(hi, i2.hi + 1); }
    lo = max(lo, max(-c.pl, c.kmin));
    hi = min(hi, min(c.tl, c.kmax));
    const bool valid = lo <= hi;
    if (tid == 0) { s_lo[s & RMASK] = valid ? lo : 1; s_hi[s & RMASK] = valid ? hi : 0; }
    if (valid) {
      int32_t* om = bs_row(c, C_M, s);
      int32_t* oi1 = bs_row(c, C_I1, s);
      int32_t* oi2 = bs_row(c, C_I2, s);
      int32_t* od1 = bs_row(c, C_D1, s);
      int32_t* od2 = bs_row(c, C_D2, s);
      int32_t* pre = c.pre + (int64_t)s * c.width;
      uint8_t* bt = c.bt + (int64_t)s * c.width;
      for (int kb = lo; kb <= hi; kb += blockDim.x) {  // uniform over the workgroup (wave_lce below)
        const int k = kb + tid;
        const bool on = k <= hi;
        const int m1a = ldk(mo1, k - 1), m1b = ldk(mo1, k + 1);
        const int m2a = ldk(mo2, k - 1), m2b = ldk(mo2, k + 1);
        const int e_i1 = ldk(i1, k - 1), e_i2 = ldk(i2, k - 1);
        const int e_d1 = ldk(d1, k + 1), e_d2 = ldk(d2, k + 1);
        unsigned bits = 0;
        // ext wins ties (WFA2-lib: ext type > open type; piggyback: ext >= open)
        if (e_i1 >= m1a) bits |= BT_I1_EXT;
        if (e_i2 >= m2a) bits |= BT_I2_EXT;
        if (e_d1 >= m1b) bits |= BT_D1_EXT;
        if (e_d2 >= m2b) bits |= BT_D2_EXT;
        int ins1 = valid_or_null(max(m1a, e_i1) + 1, k, upl, utl);
        int ins2 = valid_or_null(max(m2a, e_i2) + 1, k, upl, utl);
        int del1 = valid_or_null(max(m1b, e_d1), k, upl, utl);
        int del2 = valid_or_null(max(m2b, e_d2), k, upl, utl);
        int mis  = valid_or_null(ldk(mx, k) + 1, k, upl, utl);
        // M source priority on equal offsets: mismatch > D2 > D1 > I2 > I1
        int m = ins1; unsigned src = C_I1;
        if (ins2 >= m) { m = ins2; src = C_I2; }
        if (del1 >= m) { m = del1; src = C_D1; }
        if (del2 >= m) { m = del2; src = C_D2; }
        if (mis >= m)  { m = mis;  src = C_M; }
        if (!on) m = WF_NULL;
        const int ext = wave_lce(c.P, c.T, m - k, m, min(c.pl - (m - k), c.tl - m), m >= 0);
        if (!on) continue;
        pre[k] = m;
        bt[k] = (uint8_t)(bits | src);
        if (m >= 0) {
          m += ext;
          if (J.endsfree) {
            const int h = m, v = m - k;
            if ((h >= c.tl && c.pl - v <= J.pef) || (v >= c.pl && c.tl - h <= J.tef)) atomicMin(&s_endk, k);
          }
        }
        if (!J.endsfree && k == k_end) {
          const int ev = J.comp_end == C_M ? m : (J.comp_end == C_I1 ? ins1 : (J.comp_end == C_I2 ? ins2 : (J.comp_end == C_D1 ? del1 : del2)));
          if (ev >= c.tl) s_done = 1;
        }
        oi1[k] = ins1; oi2[k] = ins2; od1[k] = del1; od2[k] = del2; om[k] = m;
      }
      cells += (uint64_t)(hi - lo + 1);
    }
    __syncthreads();
    done = J.endsfree ? (s_endk != INT32_MAX) : (s_done != 0);
  }

  // ---- backtrace (wavefront_backtrace_affine): the first wave, every lane with the same state.  Inside a gap the walk
  // visits one cell per base and each visit is a dependent load of a decision byte; a patch begins with the ~1 kb end gap
  // of its record, so those walks were most of this kernel's time.  The cells of a gap lie on a known line -- (score - j e,
  // diagonal +- j) -- so the 64 lanes read the next 64 decision bytes at once and the walk jumps to the first one that
  // does not say "extension".  Lane 0 writes.
  if (tid < 64) {
    const int lane = tid;
    BaseResult r; r.status = status; r.score = s; r.nruns = 0; r.cells = cells;
    if (status == 0) {
      RleWriter w; w.base = rle + J.rle_end; w.n = 0; w.cur_op = -1; w.cur_len = 0; w.writes = lane == 0;
      int comp = J.endsfree ? C_M : J.comp_end;
      int k = J.endsfree ? s_endk : k_end;
      int off = J.endsfree ? bs_row(c, C_M, s)[k] : c.tl;
      int sc = s;
      int h = off, v = off - k;
      if (comp == C_M) {
        if (v < c.pl) w.push(OP_D, c.pl - v);
        if (h < c.tl) w.push(OP_I, c.tl - h);
      }
      const DevPen& pn = c.pen;
      while (v > 0 && h > 0 && sc > 0) {
        if (comp != C_M) {
          // a run of gap cells: j-th cell of the line, with the loop's own conditions
          const bool ins = comp == C_I1 || comp == C_I2;
          const int e = (comp == C_I1 || comp == C_D1) ? pn.e1 : pn.e2, o = (comp == C_I1 || comp == C_D1) ? pn.o1 : pn.o2;
          const unsigned mask = comp == C_I1 ? BT_I1_EXT : (comp == C_I2 ? BT_I2_EXT : (comp == C_D1 ? BT_D1_EXT : BT_D2_EXT));
          const int scj = sc - lane * e, kj = ins ? k - lane : k + lane;
          const bool alive = scj > 0 && (ins ? h - lane > 0 : v - lane > 0);
          const unsigned bj = alive ? c.bt[(int64_t)scj * c.width + kj] : 0u;
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
            // the cell that opened the gap
            sc -= o + e; comp = C_M;
            w.push(ins ? OP_I : OP_D, 1);
            if (ins) { --k; --off; } else ++k;
            v = off - k; h = off;
          }
          continue;
        }
        // (round 6) A run of mismatches stays on its diagonal, x scores apart -- and between unrelated sequences (the patches that pass every
        // score budget) that is what a path is made of: thousands of cells, each a dependent load.  The 64 lanes read the decision byte and the
        // offset of the next 64 cells of that line at once; the walk goes through them from registers for as long as each one's source is the
        // mismatch.  (Measured: it is NOT what C2's 9.3 ms launches of this kernel are made of -- nor are the ring loads or the extension's
        // round trips to L2, both tried in LDS / batched four diagonals at a time and taken out again: ~4000 score steps at 2.3 us, ~250
        // instructions per wave and step over rows of 3.4 k diagonals.  DESIGN.md section 8.)
        const int scj = sc - lane * pn.x;
        const unsigned bj = scj > 0 ? (unsigned)c.bt[(int64_t)scj * c.width + k] : 0u;
        const int pj = scj > 0 ? c.pre[(int64_t)scj * c.width + k] : 0;
        bool stop = false;
        for (int j = 0; j < 64; ++j) {
          const unsigned b = (unsigned)rdlane((int)bj, j);
          const int pre = rdlane(pj, j);
          w.push(OP_M, off - pre);
          off = pre; v = off - k; h = off;
          if (v <= 0 || h <= 0) { stop = true; break; }
          const unsigned src = b & 7u;
          if (src == C_M) {
            sc -= pn.x; comp = C_M; w.push(OP_X, 1); --off;
            v = off - k; h = off;
            if (!(v > 0 && h > 0 && sc > 0)) break;  // (the walk's own condition: it ends here)
            continue;                                // the next cell of the line: lane j + 1 holds it
          }
          if (src == C_I1) { if (b & BT_I1_EXT) { sc -= pn.e1; comp = C_I1; } else { sc -= pn.o1 + pn.e1; comp = C_M; } w.push(OP_I, 1); --k; --off; }
          else if (src == C_I2) { if (b & BT_I2_EXT) { sc -= pn.e2; comp = C_I2; } else { sc -= pn.o2 + pn.e2; comp = C_M; } w.push(OP_I, 1); --k; --off; }
          else if (src == C_D1) { if (b & BT_D1_EXT) { sc -= pn.e1; comp = C_D1; } else { sc -= pn.o1 + pn.e1; comp = C_M; } w.push(OP_D, 1); ++k; }
          else { if (b & BT_D2_EXT) { sc -= pn.e2; comp = C_D2; } else { sc -= pn.o2 + pn.e2; comp = C_M; } w.push(OP_D, 1); ++k; }
          v = off - k; h = off;
          break;  // the path leaves the line
        }
        if (stop) break;
      }
      if (comp == C_M && v > 0 && h > 0) { const int nm = min(v, h); w.push(OP_M, nm); v -= nm; h -= nm; }
      if (v > 0) w.push(OP_D, v);
      if (h > 0) w.push(OP_I, h);
      w.flush();
      r.nruns = w.n;
    }
    if (lane == 0) results[blockIdx.x] = r;
  }
}


// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Time-tiled phase 1 (see TileJob in wfa_device.h)
// ---------------------------------------------------------------------------
// unconditional LDS read, range applied by select (every local column k-1..k+1 is inside the tile row)


__global__ void wfa_tile_init_kernel(const uint8_t* __restrict__ seq, int32_t* __restrict__ ring_arena,
                                     const TileJob* __restrict__ jobs, int32_t* __restrict__ mak0, int njobs) {
  const int i = blockIdx.x;  // one wave per (job, direction)
  if (i >= njobs * 2) return;
  const TileJob J = jobs[i >> 1];
  const int d = i & 1;
  BpCtx c;
  c.P[0] = seq + J.p_fwd; c.T[0] = seq + J.t_fwd;
  c.P[1] = seq + J.p_rev; c.T[1] = seq + J.t_rev;
  c.width = J.width; c.koff = J.koff;
  c.ring = ring_arena + J.ring_in + c.koff;
  c.pl = J.pl; c.tl = J.tl;
  c.kb_lo = -SUB_NONE; c.kb_hi = SUB_NONE;  // rows 0: a handful of cells
  int mak = 0;
  const int end = bp_init_row0(c, d, d == 0 ? J.comp_begin : J.comp_end, d == 0 ? J.comp_end : J.comp_begin, mak);
  if (threadIdx.x == 0) {
    mak0[i * 2 + 0] = mak;
    mak0[i * 2 + 1] = end;
  }
}

// LDS rows: M[scope][Wt], I1[e1+1][Wt], D1[e1+1][Wt], I2[e2+1][Wt], D2[e2+1][Wt], mak[T+1]
__global__ __launch_bounds__(1024) void wfa_tile_kernel(const uint8_t* __restrict__ seq, int32_t* __restrict__ ring_arena,
                                                       const TileJob* __restrict__ jobs, const TileTask* __restrict__ tasks,
                                                       int32_t* __restrict__ mak_out, int T, int Wt, DevPen pen, int scope) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  TileTask tk = tasks[blockIdx.x];
  const TileJob J = jobs[tk.job];
  if (!J.active) return;
  const Rng RG = make_rng(J.pl, J.tl, J.sub);
  {  // tasks carry (tile index, tile width): this block's diagonal range [-s1, s1], clipped to the problem, is cut
     // into tiles from its own left end, so every tile but the last is full
    const int s1 = J.s0 + T;
    int L, R;
    rng_block(RG, J.s0, s1, L, R);
    const int idx = tk.core_lo, core = tk.core_hi;
    tk.core_lo = L + idx * core;
    tk.core_hi = min(R, tk.core_lo + core - 1);
    if (tk.core_lo > R) return;
  }
  const int dir = tk.dir, tid = threadIdx.x, NT = blockDim.x;
  const int n1 = pen.e1 + 1, n2 = pen.e2 + 1;
  int* sM = lds;
  int* sI1 = sM + scope * Wt;
  int* sD1 = sI1 + n1 * Wt;
  int* sI2 = sD1 + n1 * Wt;
  int* sD2 = sI2 + n2 * Wt;
  int* sMak = sD2 + n2 * Wt;
  const uint8_t* P = seq + (dir == 0 ? J.p_fwd : J.p_rev);
  const uint8_t* Tx = seq + (dir == 0 ? J.t_fwd : J.t_rev);
  const int pl = J.pl, tl = J.tl, s0 = J.s0;
  const unsigned upl = (unsigned)pl, utl = (unsigned)tl;
  const int kA = tk.core_lo - T;  // diagonal of local column 0
  const int64_t width = J.width;
  const int32_t* rin = ring_arena + J.ring_in + J.koff + (int64_t)dir * 5 * RING * width;
  int32_t* rout = ring_arena + J.ring_out + J.koff + (int64_t)dir * 5 * RING * width;
  const int ncol = tk.core_hi + T - kA + 1;  // <= Wt

  // ---- load the snapshot (rows <= s0): row loop uniform, columns across threads ----
  for (int row = 0; row < scope + 2 * pen.e1 + 2 * pen.e2; ++row) {
    int comp, sc, slot, r = row;
    int* dst;
    if (r < scope) { comp = C_M; sc = s0 - r; slot = ((sc % scope) + scope) % scope; dst = sM; }
    else if ((r -= scope) < pen.e1) { comp = C_I1; sc = s0 - r; slot = ((sc % n1) + n1) % n1; dst = sI1; }
    else if ((r -= pen.e1) < pen.e1) { comp = C_D1; sc = s0 - r; slot = ((sc % n1) + n1) % n1; dst = sD1; }
    else if ((r -= pen.e1) < pen.e2) { comp = C_I2; sc = s0 - r; slot = ((sc % n2) + n2) % n2; dst = sI2; }
    else { r -= pen.e2; comp = C_D2; sc = s0 - r; slot = ((sc % n2) + n2) % n2; dst = sD2; }
    const int lo = sc >= 0 ? rng_lo(RG, sc) : 1, hi = sc >= 0 ? rng_hi(RG, sc) : 0;
    const int32_t* src = rin + ((int64_t)(comp * RING + (sc & RMASK))) * width;
    for (int j = tid; j < Wt; j += NT) {
      const int k = kA + j;
      dst[slot * Wt + j] = (j < ncol && k >= lo && k <= hi) ? src[k] : WF_NULL;
    }
  }
  for (int t = tid; t <= T; t += NT) sMak[t] = 0;
  __syncthreads();
#ifdef WFM_PROFILE_SECTIONS
  long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_start = clock64();
#endif

  // ---- T steps inside LDS ----
  int curM = ((s0 % scope) + scope) % scope, cur1 = ((s0 % n1) + n1) % n1, cur2 = ((s0 % n2) + n2) % n2;
  for (int t = 1; t <= T; ++t) {
    SEC_T(tz);
    const int s = s0 + t;
    curM = curM + 1 == scope ? 0 : curM + 1;
    cur1 = cur1 + 1 == n1 ? 0 : cur1 + 1;
    cur2 = cur2 + 1 == n2 ? 0 : cur2 + 1;
    const int sx = s - pen.x, so1 = s - pen.o1 - pen.e1, so2 = s - pen.o2 - pen.e2, se1 = s - pen.e1, se2 = s - pen.e2;
    // closed-form ranges of the source rows (dead rows: lo > hi)
    const int lx = sx >= 0 ? rng_lo(RG, sx) : 1, hx = sx >= 0 ? rng_hi(RG, sx) : 0;
    const int l1 = so1 >= 0 ? rng_lo(RG, so1) : 1, h1 = so1 >= 0 ? rng_hi(RG, so1) : 0;
    const int l2 = so2 >= 0 ? rng_lo(RG, so2) : 1, h2 = so2 >= 0 ? rng_hi(RG, so2) : 0;
    const int le1 = se1 >= 0 ? rng_lo(RG, se1) : 1, he1 = se1 >= 0 ? rng_hi(RG, se1) : 0;
    const int le2 = se2 >= 0 ? rng_lo(RG, se2) : 1, he2 = se2 >= 0 ? rng_hi(RG, se2) : 0;
    // ring slots: (s - d) mod n == cur - d (+ n if negative), all look-backs d < n
    int qx = curM - pen.x; if (qx < 0) qx += scope;
    int q1 = curM - pen.o1 - pen.e1; if (q1 < 0) q1 += scope;
    int q2 = curM - pen.o2 - pen.e2; if (q2 < 0) q2 += scope;
    int qe1 = cur1 - pen.e1; if (qe1 < 0) qe1 += n1;
    int qe2 = cur2 - pen.e2; if (qe2 < 0) qe2 += n2;
    const int* mX = sM + qx * Wt - kA;
    const int* mO1 = sM + q1 * Wt - kA;
    const int* mO2 = sM + q2 * Wt - kA;
    const int* rI1 = sI1 + qe1 * Wt - kA;
    const int* rD1 = sD1 + qe1 * Wt - kA;
    const int* rI2 = sI2 + qe2 * Wt - kA;
    const int* rD2 = sD2 + qe2 * Wt - kA;
    int* oM = sM + curM * Wt - kA;
    int* oI1 = sI1 + cur1 * Wt - kA;
    int* oD1 = sD1 + cur1 * Wt - kA;
    int* oI2 = sI2 + cur2 * Wt - kA;
    int* oD2 = sD2 + cur2 * Wt - kA;
    const int klo = max(kA + t, rng_lo(RG, s)), khi = min(tk.core_hi + T - t, rng_hi(RG, s));
    const bool stream = (t > T - scope);  // the last `scope` rows of I/D go to the output snapshot
    int32_t* gI1 = rout + ((int64_t)(C_I1 * RING + (s & RMASK))) * width;
    int32_t* gI2 = rout + ((int64_t)(C_I2 * RING + (s & RMASK))) * width;
    int32_t* gD1 = rout + ((int64_t)(C_D1 * RING + (s & RMASK))) * width;
    int32_t* gD2 = rout + ((int64_t)(C_D2 * RING + (s & RMASK))) * width;
    int mak = 0;
    SEC_T(ta);
    SEC_ADD(4, tz, ta);
    for (int k = klo + tid; k <= khi; k += NT) {
      SEC_T(t0);
#define LDSV(p, kk, lo_, hi_) sel_rng((p)[kk], kk, lo_, hi_)
      const int m1a = LDSV(mO1, k - 1, l1, h1), m1b = LDSV(mO1, k + 1, l1, h1);
      const int m2a = LDSV(mO2, k - 1, l2, h2), m2b = LDSV(mO2, k + 1, l2, h2);
      int ins1 = max(m1a, LDSV(rI1, k - 1, le1, he1)) + 1;
      int ins2 = max(m2a, LDSV(rI2, k - 1, le2, he2)) + 1;
      int del1 = max(m1b, LDSV(rD1, k + 1, le1, he1));
      int del2 = max(m2b, LDSV(rD2, k + 1, le2, he2));
      int mis = LDSV(mX, k, lx, hx) + 1;
#undef LDSV
      ins1 = valid_or_null(ins1, k, upl, utl);
      ins2 = valid_or_null(ins2, k, upl, utl);
      del1 = valid_or_null(del1, k, upl, utl);
      del2 = valid_or_null(del2, k, upl, utl);
      mis = valid_or_null(mis, k, upl, utl);
      int m = max(imax3(ins1, ins2, mis), max(del1, del2));
      asm volatile("" :: "v"(m));
      SEC_T(t1);
      if (m >= 0) {
        m += lce_bounded2(P, Tx, m - k, m, pl, tl);
        mak = max(mak, 2 * m - k);
      }
      asm volatile("" :: "v"(m));
      SEC_T(t2);
      SEC_ADD(0, t0, t1); SEC_ADD(1, t1, t2);
      oI1[k] = ins1; oI2[k] = ins2; oD1[k] = del1; oD2[k] = del2; oM[k] = m;
      if (stream && k >= tk.core_lo && k <= tk.core_hi) { gI1[k] = ins1; gI2[k] = ins2; gD1[k] = del1; gD2[k] = del2; }
      SEC_T(t3);
      SEC_ADD(5, t2, t3);
    }
    SEC_T(td);
    mak = wave_max_dpp63(mak);
    if ((tid & 63) == 63 && mak > 0) atomicMax(&sMak[t], mak);
    SEC_T(tb);
    __syncthreads();
    SEC_T(tc);
    SEC_ADD(2, ta, tb); SEC_ADD(3, tb, tc); SEC_ADD(6, td, tb);
  }
#ifdef WFM_PROFILE_SECTIONS
  if (tid == 256 && blockIdx.x == gridDim.x / 2) { for (int q = 0; q < 7; ++q) g_sec[q] = sec[q]; g_sec[7] = clock64() - t_start; }
#endif
  // ---- write the last `scope` M rows of the core, and the per-step antidiagonal maxima ----
  const int ncore = tk.core_hi - tk.core_lo + 1;
  for (int idx = tid; idx < scope * ncore; idx += NT) {
    const int r = idx / ncore;
    const int k = tk.core_lo + (idx - r * ncore);
    const int sc = s0 + T - r;
    if (sc < 0 || k < rng_lo(RG, sc) || k > rng_hi(RG, sc)) continue;
    rout[((int64_t)(C_M * RING + (sc & RMASK))) * width + k] = sM[(sc % scope) * Wt + (k - kA)];
  }
  int32_t* mk = mak_out + ((int64_t)tk.job * 2 + dir) * T;
  for (int t = 1 + tid; t <= T; t += NT) if (sMak[t] > 0) atomicMax(&mk[t - 1], sMak[t]);
}

