/*
 * oracle/wfa2p.c -- TEST INFRASTRUCTURE ONLY.  See wfa2p.h for scope, the
 * reference call sites this restates, and the "parity unpinned" statement.
 *
 * Conventions (WFA2-lib; corroborated in-tree by wfa_edit_callback.cpp:77-116):
 *   pattern indexes v (rows), text indexes h (columns), diagonal k = h - v,
 *   a wavefront stores offset = h per diagonal, v = offset - k.
 *   'I' advances h only (k-1 -> k, offset+1); 'D' advances v only (k+1 -> k).
 *   NULL offset = INT32_MIN/2.
 *
 * Recurrences (gap-affine 2 pieces; WFA2-lib wavefront_compute_affine2p_idm):
 *   I1[s][k] = max(M[s-o1-e1][k-1], I1[s-e1][k-1]) + 1
 *   I2[s][k] = max(M[s-o2-e2][k-1], I2[s-e2][k-1]) + 1
 *   D1[s][k] = max(M[s-o1-e1][k+1], D1[s-e1][k+1])
 *   D2[s][k] = max(M[s-o2-e2][k+1], D2[s-e2][k+1])
 *   M [s][k] = max(M[s-x][k]+1, I1, I2, D1, D2); nulled if h>tlen or v>plen
 * then every wavefront is trimmed at both ends past out-of-bounds cells
 * (wavefront_compute_trim_ends) and M is extended along exact matches.
 *
 * Backtrace tie-break (WFA2-lib backtrace_type order): among predecessors of
 * equal offset  mismatch > D2_ext > D2_open > D1_ext > D1_open > I2_ext >
 * I2_open > I1_ext > I1_open.  The piggy-back compute kernel (MemoryMed) makes
 * the same choices cell by cell (sequential ifs ins1, ins2, del1, del2, misms;
 * ext >= open), so one policy serves MemoryHigh, MemoryMed and the BiWFA base.
 */
#include "wfa2p.h"

#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WF_NULL (INT32_MIN / 2)
#define MAXI(a, b) ((a) > (b) ? (a) : (b))
#define MINI(a, b) ((a) < (b) ? (a) : (b))

/* BiWFA constants (WFA2-lib wavefront_bialign.c) */
#define BIALIGN_FALLBACK_MIN_SCORE 250
#define BIALIGN_FALLBACK_MIN_LENGTH 100

#define ST_OK 0
#define ST_END_REACHED 1
#define ST_UNREACHABLE (-300)
#define ST_OOM (-200)

typedef struct {
  int lo, hi;     /* effective range */
  int clo, chi;   /* allocated range */
  int32_t* mem;
  int32_t* off;   /* off[k] valid for clo<=k<=chi */
} wf_t;

typedef struct {
  char *pbuf, *tbuf;      /* padded copies */
  const char *p, *t;
  int plen, tlen;
  wfo_penalties_t pen;
  int scope;              /* max(x, o1+e1, o2+e2) + 1 */
  int modular;
  int nslots;
  wf_t** wf[5];
  int comp_begin, comp_end;
  int endsfree, pbf, pef, tbf, tef;
  int num_null_steps;
  int end_score, end_k, end_off;
  int32_t* nullrow_mem;
  int32_t* nullrow;       /* valid for -plen-4 .. tlen+4 */
  wfo_stats_t* st;
  int bounded, sub;       /* wfo_find_breakpoint_bounded: rows only hold |k - (tlen - plen)| <= sub - s */
} al_t;

/* ------------------------------------------------------------------ */
static wf_t* wf_new(int lo, int hi) {
  wf_t* w = (wf_t*)malloc(sizeof(wf_t));
  if (!w) return NULL;
  w->lo = lo; w->hi = hi; w->clo = lo; w->chi = hi;
  size_t n = (size_t)(hi - lo + 1);
  w->mem = (int32_t*)malloc(n * sizeof(int32_t) + 8);
  if (!w->mem) { free(w); return NULL; }
  w->off = w->mem - lo;
  return w;
}
static void wf_free(wf_t* w) { if (w) { free(w->mem); free(w); } }

static inline int32_t wf_get(const wf_t* w, int k) {
  return (w && k >= w->lo && k <= w->hi) ? w->off[k] : WF_NULL;
}

static int al_init(al_t* a, const char* p, int plen, const char* t, int tlen,
                   const wfo_penalties_t* pen, int modular, int reverse, wfo_stats_t* st) {
  memset(a, 0, sizeof(*a));
  a->plen = plen; a->tlen = tlen; a->pen = *pen; a->st = st;
  a->pbuf = (char*)malloc((size_t)plen + 48);
  a->tbuf = (char*)malloc((size_t)tlen + 48);
  if (!a->pbuf || !a->tbuf) return ST_OOM;
  /* sentinels differ between pattern and text so extend needs no bound test */
  memset(a->pbuf, '?', (size_t)plen + 48);
  memset(a->tbuf, '!', (size_t)tlen + 48);
  if (!reverse) {
    memcpy(a->pbuf + 16, p, (size_t)plen);
    memcpy(a->tbuf + 16, t, (size_t)tlen);
  } else {
    for (int i = 0; i < plen; ++i) a->pbuf[16 + i] = p[plen - 1 - i];
    for (int i = 0; i < tlen; ++i) a->tbuf[16 + i] = t[tlen - 1 - i];
  }
  a->p = a->pbuf + 16; a->t = a->tbuf + 16;
  int sc = MAXI(pen->x, MAXI(pen->o1 + pen->e1, pen->o2 + pen->e2)) + 1;
  a->scope = sc;
  a->modular = modular;
  a->nslots = modular ? sc : 64;
  for (int c = 0; c < 5; ++c) {
    a->wf[c] = (wf_t**)calloc((size_t)a->nslots, sizeof(wf_t*));
    if (!a->wf[c]) return ST_OOM;
  }
  size_t nn = (size_t)plen + (size_t)tlen + 16;
  a->nullrow_mem = (int32_t*)malloc(nn * sizeof(int32_t));
  if (!a->nullrow_mem) return ST_OOM;
  for (size_t i = 0; i < nn; ++i) a->nullrow_mem[i] = WF_NULL;
  a->nullrow = a->nullrow_mem + plen + 6;
  return ST_OK;
}

static void al_free(al_t* a) {
  for (int c = 0; c < 5; ++c) {
    if (a->wf[c]) {
      for (int i = 0; i < a->nslots; ++i) wf_free(a->wf[c][i]);
      free(a->wf[c]);
    }
  }
  free(a->pbuf); free(a->tbuf); free(a->nullrow_mem);
}

static inline int slot_of(const al_t* a, int s) { return a->modular ? (s % a->scope) : s; }

static int ensure_slots(al_t* a, int s) {
  if (a->modular || s < a->nslots) return ST_OK;
  int n = a->nslots;
  while (n <= s) n *= 2;
  for (int c = 0; c < 5; ++c) {
    wf_t** nw = (wf_t**)realloc(a->wf[c], (size_t)n * sizeof(wf_t*));
    if (!nw) return ST_OOM;
    memset(nw + a->nslots, 0, (size_t)(n - a->nslots) * sizeof(wf_t*));
    a->wf[c] = nw;
  }
  a->nslots = n;
  return ST_OK;
}

static inline wf_t* get_wf(const al_t* a, int c, int s) {
  if (s < 0) return NULL;
  if (!a->modular && s >= a->nslots) return NULL;
  return a->wf[c][slot_of(a, s)];
}

static void set_wf(al_t* a, int c, int s, wf_t* w) {
  int sl = slot_of(a, s);
  if (a->wf[c][sl]) wf_free(a->wf[c][sl]);
  a->wf[c][sl] = w;
}

/* wavefront_compute_trim_ends */
static void trim_ends(const al_t* a, wf_t* w) {
  const uint32_t plen = (uint32_t)a->plen, tlen = (uint32_t)a->tlen;
  int k;
  for (k = w->hi; k >= w->lo; --k) {
    int32_t o = w->off[k];
    uint32_t h = (uint32_t)o, v = (uint32_t)(o - k);
    if (h <= tlen && v <= plen) break;
  }
  w->hi = k;
  for (k = w->lo; k <= w->hi; ++k) {
    int32_t o = w->off[k];
    uint32_t h = (uint32_t)o, v = (uint32_t)(o - k);
    if (h <= tlen && v <= plen) break;
  }
  w->lo = k;
}

/* longest common extension from (v,h); buffers are sentinel padded */
static inline int lce(const char* p, const char* t) {
  int n = 0;
  for (;;) {
    uint64_t a, b;
    memcpy(&a, p + n, 8);
    memcpy(&b, t + n, 8);
    uint64_t x = a ^ b;
    if (x) return n + (__builtin_ctzll(x) >> 3);
    n += 8;
  }
}

/* wavefront_compute_affine2p (+ limits_input, allocate_output, trim) */
static int compute_step(al_t* a, int s) {
  const wfo_penalties_t* pn = &a->pen;
  if (ensure_slots(a, s) != ST_OK) return ST_OOM;
  const wf_t* m_x  = get_wf(a, WFO_M,  s - pn->x);
  const wf_t* m_o1 = get_wf(a, WFO_M,  s - pn->o1 - pn->e1);
  const wf_t* m_o2 = get_wf(a, WFO_M,  s - pn->o2 - pn->e2);
  const wf_t* i1e  = get_wf(a, WFO_I1, s - pn->e1);
  const wf_t* i2e  = get_wf(a, WFO_I2, s - pn->e2);
  const wf_t* d1e  = get_wf(a, WFO_D1, s - pn->e1);
  const wf_t* d2e  = get_wf(a, WFO_D2, s - pn->e2);
  /* a wavefront whose range was trimmed to empty counts as null */
  #define ISNULL(w) (!(w) || (w)->lo > (w)->hi)
  if (ISNULL(m_x)) m_x = NULL;
  if (ISNULL(m_o1)) m_o1 = NULL;
  if (ISNULL(m_o2)) m_o2 = NULL;
  if (ISNULL(i1e)) i1e = NULL;
  if (ISNULL(i2e)) i2e = NULL;
  if (ISNULL(d1e)) d1e = NULL;
  if (ISNULL(d2e)) d2e = NULL;
  if (!m_x && !m_o1 && !m_o2 && !i1e && !i2e && !d1e && !d2e) {
    a->num_null_steps++;
    for (int c = 0; c < 5; ++c) set_wf(a, c, s, NULL);
    return ST_OK;
  }
  a->num_null_steps = 0;
  /* limits (wavefront_compute_limits_input) */
  int lo = INT_MAX, hi = INT_MIN;
  if (m_x)  { lo = MINI(lo, m_x->lo);      hi = MAXI(hi, m_x->hi); }
  if (m_o1) { lo = MINI(lo, m_o1->lo - 1); hi = MAXI(hi, m_o1->hi + 1); }
  if (m_o2) { lo = MINI(lo, m_o2->lo - 1); hi = MAXI(hi, m_o2->hi + 1); }
  if (i1e)  { lo = MINI(lo, i1e->lo + 1);  hi = MAXI(hi, i1e->hi + 1); }
  if (i2e)  { lo = MINI(lo, i2e->lo + 1);  hi = MAXI(hi, i2e->hi + 1); }
  if (d1e)  { lo = MINI(lo, d1e->lo - 1);  hi = MAXI(hi, d1e->hi - 1); }
  if (d2e)  { lo = MINI(lo, d2e->lo - 1);  hi = MAXI(hi, d2e->hi - 1); }
  if (a->bounded) {
    /* The product's score bound (wfmash_amd/csrc/wfa_kernels.hip, struct Rng): a cell from which the end diagonal is out of
     * reach within the bound is not computed.  Not part of the reference's algorithm: this branch exists so that the claim
     * "the bound does not change the breakpoint" can be checked on the CPU (tests/test_oracle_wfa.py). */
    const int kinv = a->tlen - a->plen;
    lo = MAXI(lo, kinv - (a->sub - s));
    hi = MINI(hi, kinv + (a->sub - s));
  }
  if (lo > hi) { /* cannot happen with a non-null input, keep defensive */
    for (int c = 0; c < 5; ++c) set_wf(a, c, s, NULL);
    return ST_OK;
  }
  /* allocate outputs (wavefront_compute_allocate_output) */
  wf_t* om  = wf_new(lo, hi);
  wf_t* oi1 = (m_o1 || i1e) ? wf_new(lo, hi) : NULL;
  wf_t* oi2 = (m_o2 || i2e) ? wf_new(lo, hi) : NULL;
  wf_t* od1 = (m_o1 || d1e) ? wf_new(lo, hi) : NULL;
  wf_t* od2 = (m_o2 || d2e) ? wf_new(lo, hi) : NULL;
  if (!om || ((m_o1 || i1e) && !oi1) || ((m_o2 || i2e) && !oi2) ||
      ((m_o1 || d1e) && !od1) || ((m_o2 || d2e) && !od2)) {
    wf_free(om); wf_free(oi1); wf_free(oi2); wf_free(od1); wf_free(od2);
    return ST_OOM;
  }
  const uint32_t plen = (uint32_t)a->plen, tlen = (uint32_t)a->tlen;
  /* interior where every present input covers k-1..k+1 -> plain pointer loop */
  int ilo = lo, ihi = hi;
  #define NARROW(w) if (w) { ilo = MAXI(ilo, (w)->lo + 1); ihi = MINI(ihi, (w)->hi - 1); }
  NARROW(m_x) NARROW(m_o1) NARROW(m_o2) NARROW(i1e) NARROW(i2e) NARROW(d1e) NARROW(d2e)
  if (ilo > ihi) { ilo = hi + 1; ihi = hi; }
  /* absent inputs read from the all-NULL row (covers -plen-4..tlen+4) */
  if (ilo < -a->plen - 3 || ihi > a->tlen + 3) { ilo = hi + 1; ihi = hi; }
  const int32_t* pm_x  = m_x  ? m_x->off  : a->nullrow;
  const int32_t* pm_o1 = m_o1 ? m_o1->off : a->nullrow;
  const int32_t* pm_o2 = m_o2 ? m_o2->off : a->nullrow;
  const int32_t* pi1   = i1e  ? i1e->off  : a->nullrow;
  const int32_t* pi2   = i2e  ? i2e->off  : a->nullrow;
  const int32_t* pd1   = d1e  ? d1e->off  : a->nullrow;
  const int32_t* pd2   = d2e  ? d2e->off  : a->nullrow;
  int32_t dummy_i1, dummy_i2, dummy_d1, dummy_d2;
  for (int k = lo; k <= hi; ++k) {
    int32_t ins1, ins2, del1, del2, misms;
    if (k >= ilo && k <= ihi) {
      ins1 = MAXI(pm_o1[k - 1], pi1[k - 1]) + 1;
      ins2 = MAXI(pm_o2[k - 1], pi2[k - 1]) + 1;
      del1 = MAXI(pm_o1[k + 1], pd1[k + 1]);
      del2 = MAXI(pm_o2[k + 1], pd2[k + 1]);
      misms = pm_x[k] + 1;
    } else {
      ins1 = MAXI(wf_get(m_o1, k - 1), wf_get(i1e, k - 1)) + 1;
      ins2 = MAXI(wf_get(m_o2, k - 1), wf_get(i2e, k - 1)) + 1;
      del1 = MAXI(wf_get(m_o1, k + 1), wf_get(d1e, k + 1));
      del2 = MAXI(wf_get(m_o2, k + 1), wf_get(d2e, k + 1));
      misms = wf_get(m_x, k) + 1;
    }
    *(oi1 ? &oi1->off[k] : &dummy_i1) = ins1;
    *(oi2 ? &oi2->off[k] : &dummy_i2) = ins2;
    *(od1 ? &od1->off[k] : &dummy_d1) = del1;
    *(od2 ? &od2->off[k] : &dummy_d2) = del2;
    int32_t mx = MAXI(MAXI(del1, del2), MAXI(misms, MAXI(ins1, ins2)));
    uint32_t h = (uint32_t)mx, v = (uint32_t)(mx - k);
    if (h > tlen) mx = WF_NULL;
    if (v > plen) mx = WF_NULL;
    om->off[k] = mx;
  }
  if (a->st) a->st->cells += (uint64_t)(hi - lo + 1);
  /* wavefront_compute_process_ends: trim all five */
  trim_ends(a, om);
  if (oi1) trim_ends(a, oi1);
  if (oi2) trim_ends(a, oi2);
  if (od1) trim_ends(a, od1);
  if (od2) trim_ends(a, od2);
  set_wf(a, WFO_M, s, om);
  set_wf(a, WFO_I1, s, oi1);
  set_wf(a, WFO_I2, s, oi2);
  set_wf(a, WFO_D1, s, od1);
  set_wf(a, WFO_D2, s, od2);
  return ST_OK;
}

/* wavefront_termination_end2end */
static int termination_end2end(al_t* a, int s) {
  const int ak = a->tlen - a->plen, aoff = a->tlen;
  const wf_t* w = get_wf(a, a->comp_end, s);
  if (!w || w->lo > ak || ak > w->hi) return 0;
  if (w->off[ak] < aoff) return 0;
  a->end_score = s; a->end_k = ak; a->end_off = aoff;
  return 1;
}

/* wavefront_termination_endsfree (per extended cell, ascending k) */
static int termination_endsfree(al_t* a, int s, int k, int32_t off) {
  const int h = off, v = off - k;
  if (h >= a->tlen) {
    if (a->plen - v <= a->pef) { a->end_score = s; a->end_k = k; a->end_off = off; return 1; }
  }
  if (v >= a->plen) {
    if (a->tlen - h <= a->tef) { a->end_score = s; a->end_k = k; a->end_off = off; return 1; }
  }
  return 0;
}

/* wavefront_extend_end2end / _endsfree / _end2end_max.
 * Returns 1 if the alignment terminated at this score, 0 otherwise,
 * <0 on unreachable.  max_ak (may be NULL) receives max antidiagonal. */
static int extend_step(al_t* a, int s, int* max_ak, int check_term) {
  wf_t* m = get_wf(a, WFO_M, s);
  if (max_ak) *max_ak = 0;
  if (!m) {
    /* WFA2-lib: a null M-wavefront skips the termination test */
    if (a->num_null_steps > a->scope) return ST_UNREACHABLE;
    return 0;
  }
  int mak = 0;
  for (int k = m->lo; k <= m->hi; ++k) {
    int32_t off = m->off[k];
    if (off < 0) continue; /* NULL (M cells are either valid or exactly NULL) */
    int v = off - k, h = off;
    int n = lce(a->p + v, a->t + h);
    /* sentinels bound the extension at plen/tlen */
    off += n;
    if (a->st) a->st->extend_bases += (uint64_t)n;
    m->off[k] = off;
    int ak = 2 * off - k;
    if (ak > mak) mak = ak;
    if (a->endsfree && check_term) {
      if (termination_endsfree(a, s, k, off)) { if (max_ak) *max_ak = mak; return 1; }
    }
  }
  if (max_ak) *max_ak = mak;
  if (!a->endsfree && check_term) return termination_end2end(a, s);
  return 0;
}

/* wavefront_unialign_init (end2end / endsfree initial wavefronts) */
static int init_wavefronts(al_t* a) {
  for (int c = 0; c < 5; ++c) set_wf(a, c, 0, NULL);
  a->num_null_steps = 0;
  if (a->endsfree) {
    int lo = -a->pbf, hi = a->tbf;
    wf_t* m = wf_new(lo, hi);
    if (!m) return ST_OOM;
    m->off[0] = 0;
    for (int h = 1; h <= a->tbf; ++h) m->off[h] = h;
    for (int v = 1; v <= a->pbf; ++v) m->off[-v] = 0;
    set_wf(a, WFO_M, 0, m);
    return ST_OK;
  }
  wf_t* w = wf_new(0, 0);
  if (!w) return ST_OOM;
  w->off[0] = 0;
  set_wf(a, a->comp_begin, 0, w);
  return ST_OK;
}

/* ------------------------------------------------------------------ */
/* backtrace (wavefront_backtrace_affine) -- writes ops reversed into rb */
typedef struct { char* buf; int n; int cap; } rbuf_t;
static inline void rb_push(rbuf_t* r, char op, int count) {
  for (int i = 0; i < count; ++i) r->buf[r->n++] = op;
}

enum { BT_I1_OPEN = 1, BT_I1_EXT, BT_I2_OPEN, BT_I2_EXT, BT_D1_OPEN, BT_D1_EXT, BT_D2_OPEN, BT_D2_EXT, BT_M };

static inline int64_t bt_pack(int32_t off, int type) {
  return off < 0 ? (int64_t)WF_NULL * 16 : (((int64_t)off << 4) | type);
}

static int backtrace(al_t* a, rbuf_t* rb) {
  const wfo_penalties_t* pn = &a->pen;
  int comp = a->endsfree ? WFO_M : a->comp_end;
  int s = a->end_score, k = a->end_k;
  int32_t off = a->end_off;
  int h = off, v = off - k;
  if (comp == WFO_M) { /* trailing free ends (ends-free only) */
    if (v < a->plen) rb_push(rb, 'D', a->plen - v);
    if (h < a->tlen) rb_push(rb, 'I', a->tlen - h);
  }
  while (v > 0 && h > 0 && s > 0) {
    const int s_x = s - pn->x, s_o1 = s - pn->o1 - pn->e1, s_o2 = s - pn->o2 - pn->e2;
    const int s_e1 = s - pn->e1, s_e2 = s - pn->e2;
    int64_t best;
    #define SRC(c, sc, kk, add, ty) bt_pack((sc) < 0 ? WF_NULL : (wf_get(get_wf(a, c, sc), kk) < 0 ? WF_NULL : wf_get(get_wf(a, c, sc), kk) + (add)), ty)
    if (comp == WFO_M) {
      int64_t c_x   = SRC(WFO_M,  s_x,  k,     1, BT_M);
      int64_t c_i1o = SRC(WFO_M,  s_o1, k - 1, 1, BT_I1_OPEN);
      int64_t c_i1e = SRC(WFO_I1, s_e1, k - 1, 1, BT_I1_EXT);
      int64_t c_i2o = SRC(WFO_M,  s_o2, k - 1, 1, BT_I2_OPEN);
      int64_t c_i2e = SRC(WFO_I2, s_e2, k - 1, 1, BT_I2_EXT);
      int64_t c_d1o = SRC(WFO_M,  s_o1, k + 1, 0, BT_D1_OPEN);
      int64_t c_d1e = SRC(WFO_D1, s_e1, k + 1, 0, BT_D1_EXT);
      int64_t c_d2o = SRC(WFO_M,  s_o2, k + 1, 0, BT_D2_OPEN);
      int64_t c_d2e = SRC(WFO_D2, s_e2, k + 1, 0, BT_D2_EXT);
      best = c_x;
      if (c_i1o > best) best = c_i1o;
      if (c_i1e > best) best = c_i1e;
      if (c_i2o > best) best = c_i2o;
      if (c_i2e > best) best = c_i2e;
      if (c_d1o > best) best = c_d1o;
      if (c_d1e > best) best = c_d1e;
      if (c_d2o > best) best = c_d2o;
      if (c_d2e > best) best = c_d2e;
    } else if (comp == WFO_I1) {
      int64_t o = SRC(WFO_M, s_o1, k - 1, 1, BT_I1_OPEN), e = SRC(WFO_I1, s_e1, k - 1, 1, BT_I1_EXT);
      best = o > e ? o : e;
    } else if (comp == WFO_I2) {
      int64_t o = SRC(WFO_M, s_o2, k - 1, 1, BT_I2_OPEN), e = SRC(WFO_I2, s_e2, k - 1, 1, BT_I2_EXT);
      best = o > e ? o : e;
    } else if (comp == WFO_D1) {
      int64_t o = SRC(WFO_M, s_o1, k + 1, 0, BT_D1_OPEN), e = SRC(WFO_D1, s_e1, k + 1, 0, BT_D1_EXT);
      best = o > e ? o : e;
    } else {
      int64_t o = SRC(WFO_M, s_o2, k + 1, 0, BT_D2_OPEN), e = SRC(WFO_D2, s_e2, k + 1, 0, BT_D2_EXT);
      best = o > e ? o : e;
    }
    #undef SRC
    if (best < 0) return ST_UNREACHABLE; /* broken chain: must not happen */
    const int32_t max_off = (int32_t)(best >> 4);
    const int type = (int)(best & 15);
    if (comp == WFO_M) {
      int nm = off - max_off;
      if (nm < 0) return ST_UNREACHABLE;
      rb_push(rb, 'M', nm);
      off = max_off;
      v = off - k; h = off;
      if (v <= 0 || h <= 0) break;
    }
    switch (type) {
      case BT_M:       s = s_x;  comp = WFO_M;  break;
      case BT_I1_OPEN: s = s_o1; comp = WFO_M;  break;
      case BT_I1_EXT:  s = s_e1; comp = WFO_I1; break;
      case BT_I2_OPEN: s = s_o2; comp = WFO_M;  break;
      case BT_I2_EXT:  s = s_e2; comp = WFO_I2; break;
      case BT_D1_OPEN: s = s_o1; comp = WFO_M;  break;
      case BT_D1_EXT:  s = s_e1; comp = WFO_D1; break;
      case BT_D2_OPEN: s = s_o2; comp = WFO_M;  break;
      case BT_D2_EXT:  s = s_e2; comp = WFO_D2; break;
      default: return ST_UNREACHABLE;
    }
    switch (type) {
      case BT_M: rb_push(rb, 'X', 1); --off; break;
      case BT_I1_OPEN: case BT_I1_EXT: case BT_I2_OPEN: case BT_I2_EXT:
        rb_push(rb, 'I', 1); --k; --off; break;
      default: rb_push(rb, 'D', 1); ++k; break;
    }
    v = off - k; h = off;
  }
  /* account for the beginning */
  if (comp == WFO_M && v > 0 && h > 0) {
    int nm = MINI(v, h);
    rb_push(rb, 'M', nm);
    v -= nm; h -= nm;
  }
  if (v > 0) rb_push(rb, 'D', v);
  if (h > 0) rb_push(rb, 'I', h);
  return ST_OK;
}

/* wavefront_unialign: extend s, compute s+1, ...  Appends forward ops. */
static int unialign(const char* p, int plen, const char* t, int tlen, const wfo_penalties_t* pen,
                    int endsfree, int pbf, int pef, int tbf, int tef,
                    int comp_begin, int comp_end,
                    char* ops_out, int* nops, int* score, wfo_stats_t* st) {
  al_t a;
  int rc = al_init(&a, p, plen, t, tlen, pen, 0, 0, st);
  if (rc != ST_OK) { al_free(&a); return rc; }
  a.endsfree = endsfree; a.pbf = pbf; a.pef = pef; a.tbf = tbf; a.tef = tef;
  a.comp_begin = comp_begin; a.comp_end = comp_end;
  rc = init_wavefronts(&a);
  if (rc != ST_OK) { al_free(&a); return rc; }
  /* generous bound: every base as a mismatch/gap */
  const int64_t max_score = (int64_t)(pen->o2 + pen->o1) * 2 + (int64_t)(plen + tlen + 2) * MAXI(pen->x, MAXI(pen->e1, pen->e2)) + 64;
  int s = 0;
  for (;;) {
    int fin = extend_step(&a, s, NULL, 1);
    if (fin < 0) { al_free(&a); return fin; }
    if (fin) break;
    ++s;
    if (s > max_score) { al_free(&a); return ST_UNREACHABLE; }
    rc = compute_step(&a, s);
    if (rc != ST_OK) { al_free(&a); return rc; }
  }
  rbuf_t rb;
  rb.cap = plen + tlen + 2; rb.n = 0;
  rb.buf = (char*)malloc((size_t)rb.cap + 8);
  if (!rb.buf) { al_free(&a); return ST_OOM; }
  rc = backtrace(&a, &rb);
  if (rc == ST_OK) {
    for (int i = 0; i < rb.n; ++i) ops_out[i] = rb.buf[rb.n - 1 - i];
    *nops = rb.n;
    *score = a.end_score;
  }
  free(rb.buf);
  al_free(&a);
  return rc;
}

/* ------------------------------------------------------------------ */
/* BiWFA (wavefront_bialign.c) */

static inline int gap_open_of(const wfo_penalties_t* pn, int comp) {
  return (comp == WFO_I1 || comp == WFO_D1) ? pn->o1 : pn->o2;
}

/* wavefront_bialign_breakpoint_indel2indel / _m2m.  wf0 belongs to the
 * direction that just advanced; fwd0 tells whether that is the forward one. */
static void bp_check(const al_t* a0, int fwd0, int s0, int s1, const wf_t* w0, const wf_t* w1,
                     int comp, wfo_breakpoint_t* bp) {
  const int tlen = a0->tlen, plen = a0->plen;
  const int gopen = comp == WFO_M ? 0 : gap_open_of(&a0->pen, comp);
  const int lo0 = w0->lo, hi0 = w0->hi;
  const int lo1 = (tlen - plen) - w1->hi, hi1 = (tlen - plen) - w1->lo;
  if (hi1 < lo0 || hi0 < lo1) return;
  const int min_hi = MINI(hi0, hi1), max_lo = MAXI(lo0, lo1);
  for (int k0 = max_lo; k0 <= min_hi; ++k0) {
    const int k1 = (tlen - plen) - k0;
    const int32_t o0 = w0->off[k0], o1 = w1->off[k1];
    if (o0 + o1 >= tlen && s0 + s1 - gopen < bp->score) {
      if (fwd0) {
        const int v = o0 - k0, h = o0;
        if (v > plen || h > tlen) continue;
        bp->score_forward = s0; bp->score_reverse = s1;
        bp->k_forward = k0; bp->k_reverse = k1;
        bp->offset_forward = o0; bp->offset_reverse = o1;
      } else {
        const int v = o1 - k1, h = o1;
        if (v > plen || h > tlen) continue;
        bp->score_forward = s1; bp->score_reverse = s0;
        bp->k_forward = k1; bp->k_reverse = k0;
        bp->offset_forward = o1; bp->offset_reverse = o0;
      }
      bp->score = s0 + s1 - gopen;
      bp->component = comp;
      return;
    }
  }
}

/* wavefront_bialign_overlap */
static void overlap(const al_t* a0, const al_t* a1, int s0, int s1, int fwd0, wfo_breakpoint_t* bp) {
  const wfo_penalties_t* pn = &a0->pen;
  const wf_t* m0 = get_wf(a0, WFO_M, s0);
  if (!m0) return;
  const wf_t* d20 = get_wf(a0, WFO_D2, s0);
  const wf_t* i20 = get_wf(a0, WFO_I2, s0);
  const wf_t* d10 = get_wf(a0, WFO_D1, s0);
  const wf_t* i10 = get_wf(a0, WFO_I1, s0);
  for (int i = 0; i < a0->scope; ++i) {
    const int si = s1 - i;
    if (si < 0) break;
    if (s0 + si - pn->o2 >= bp->score) continue;
    const wf_t* w;
    w = get_wf(a1, WFO_D2, si); if (d20 && w) bp_check(a0, fwd0, s0, si, d20, w, WFO_D2, bp);
    w = get_wf(a1, WFO_I2, si); if (i20 && w) bp_check(a0, fwd0, s0, si, i20, w, WFO_I2, bp);
    if (s0 + si - pn->o1 >= bp->score) continue;
    w = get_wf(a1, WFO_D1, si); if (d10 && w) bp_check(a0, fwd0, s0, si, d10, w, WFO_D1, bp);
    w = get_wf(a1, WFO_I1, si); if (i10 && w) bp_check(a0, fwd0, s0, si, i10, w, WFO_I1, bp);
    if (s0 + si >= bp->score) continue;
    w = get_wf(a1, WFO_M, si); if (w) bp_check(a0, fwd0, s0, si, m0, w, WFO_M, bp);
  }
}

/* wavefront_bialign_find_breakpoint */
static int find_breakpoint_sub(const char* p, int plen, const char* t, int tlen, const wfo_penalties_t* pen,
                               int comp_begin, int comp_end, wfo_breakpoint_t* bp, wfo_stats_t* st, int bounded, int sub,
                               int tests_per_round, int* rounds_out);
static int find_breakpoint(const char* p, int plen, const char* t, int tlen, const wfo_penalties_t* pen,
                           int comp_begin, int comp_end, wfo_breakpoint_t* bp, wfo_stats_t* st) {
  return find_breakpoint_sub(p, plen, t, tlen, pen, comp_begin, comp_end, bp, st, 0, 0, 0, NULL);
}
/* bounded != 0: the product's form of the search under an upper bound `sub` of the score -- rows cut as above, and the
 * second loop starts as if a breakpoint of score sub + 1 were in hand.  Returns ST_OK with bp->score <= sub, or
 * ST_UNREACHABLE when nothing lies within the bound. */
/* tests_per_round > 0: the product's phase 2 in rounds (wfa_host.hip, run_p2_phase / wfa_p2_overlap_kernel): after that many
 * tests the walk is cut -- what it has taken so far is set aside, the next round starts "as if a breakpoint of that score
 * were in hand" with an empty record of its own, and when the loop ends in a round that took nothing the one set aside
 * stands (WFM_DEV_P2_NOTHING). */
static int find_breakpoint_sub(const char* p, int plen, const char* t, int tlen, const wfo_penalties_t* pen,
                               int comp_begin, int comp_end, wfo_breakpoint_t* bp, wfo_stats_t* st, int bounded, int sub,
                               int tests_per_round, int* rounds_out) {
  al_t f, r;
  wfo_breakpoint_t carry;
  int have_carry = 0, tests = 0, rounds = 1;
  int rc = al_init(&f, p, plen, t, tlen, pen, 1, 0, st);
  if (rc != ST_OK) { al_free(&f); return rc; }
  rc = al_init(&r, p, plen, t, tlen, pen, 1, 1, st);
  if (rc != ST_OK) { al_free(&f); al_free(&r); return rc; }
  f.comp_begin = comp_begin; f.comp_end = comp_end;
  r.comp_begin = comp_end;   r.comp_end = comp_begin;
  f.bounded = r.bounded = bounded; f.sub = r.sub = sub;
  rc = init_wavefronts(&f); if (rc == ST_OK) rc = init_wavefronts(&r);
  if (rc != ST_OK) { al_free(&f); al_free(&r); return rc; }
  const int max_antidiag = plen + tlen - 1;
  const int64_t max_steps = (int64_t)(pen->o2 + pen->o1) * 4 + (int64_t)(plen + tlen + 2) * MAXI(pen->x, MAXI(pen->e1, pen->e2)) * 2 + 256;
  int sf = 0, sr = 0, fmax = 0, rmax = 0, mak = 0, fin;
  bp->score = bounded ? sub + 1 : INT_MAX;
  bp->component = -1;
  fin = extend_step(&f, sf, &fmax, 1);
  if (fin == 1) { al_free(&f); al_free(&r); return ST_END_REACHED; }
  fin = extend_step(&r, sr, &rmax, 1);
  if (fin == 1) { al_free(&f); al_free(&r); return ST_END_REACHED; }
  int last_fwd = 0;
  rc = ST_OK;
  for (;;) {
    if (fmax + rmax >= max_antidiag) break;
    ++sf;
    if ((rc = compute_step(&f, sf)) != ST_OK) goto done;
    extend_step(&f, sf, &mak, 1);
    if (fmax < mak) fmax = mak;
    last_fwd = 1;
    if (fmax + rmax >= max_antidiag) break;
    ++sr;
    if ((rc = compute_step(&r, sr)) != ST_OK) goto done;
    extend_step(&r, sr, &mak, 1);
    if (rmax < mak) rmax = mak;
    last_fwd = 0;
    if ((int64_t)sf + sr > max_steps) { rc = ST_UNREACHABLE; goto done; }
    if (bounded && 2 * sf > sub + 128) { rc = ST_UNREACHABLE; goto done; }  /* the product gives up here too */
  }
  {
    const int scope = f.scope;
    const int gopen = MAXI(pen->o1, pen->o2);
    for (;;) {
#define WFO_ROUND_CUT()                                                                                   \
  do {                                                                                                     \
    if (tests_per_round > 0 && tests > 0 && tests % tests_per_round == 0) {                                \
      if (bp->component >= 0) { carry = *bp; have_carry = 1; }                                             \
      const int seed = bp->score;                                                                          \
      memset(bp, 0, sizeof(*bp)); bp->score = seed; bp->component = -1;  /* a record of the round's own */ \
      ++rounds;                                                                                            \
    }                                                                                                      \
  } while (0)
      if (last_fwd) {
        const int min_sr = (sr > scope - 1) ? sr - (scope - 1) : 0;
        if (sf + min_sr - gopen >= bp->score) break;
        WFO_ROUND_CUT();
        overlap(&f, &r, sf, sr, 1, bp);
        ++tests;
        ++sr;
        if ((rc = compute_step(&r, sr)) != ST_OK) goto done;
        extend_step(&r, sr, NULL, 1);
      }
      const int min_sf = (sf > scope - 1) ? sf - (scope - 1) : 0;
      if (min_sf + sr - gopen >= bp->score) break;
      WFO_ROUND_CUT();
      overlap(&r, &f, sr, sf, 0, bp);
      ++tests;
      ++sf;
      if ((rc = compute_step(&f, sf)) != ST_OK) goto done;
      extend_step(&f, sf, NULL, 1);
      if ((int64_t)sf + sr > max_steps && bp->score == INT_MAX) { rc = ST_UNREACHABLE; goto done; }
      last_fwd = 1;
    }
  }
  if (bp->component < 0 && have_carry) *bp = carry;       /* the last round took nothing: the one set aside stands */
  if (rounds_out) *rounds_out = rounds;
  if (bounded && bp->component < 0) rc = ST_UNREACHABLE;  /* the loop ended on the stand-in, no breakpoint was taken */
done:
  al_free(&f); al_free(&r);
  return rc;
}

typedef struct { char* buf; int64_t n; } fbuf_t;

static int bialign_rec(const char* p, int plen, const char* t, int tlen, const wfo_penalties_t* pen,
                       int comp_begin, int comp_end, int score_remaining, fbuf_t* out,
                       int level, wfo_stats_t* st) {
  if (st && (uint32_t)level > st->max_depth) st->max_depth = (uint32_t)level;
  if (tlen == 0) { memset(out->buf + out->n, 'D', (size_t)plen); out->n += plen; return ST_OK; }
  if (plen == 0) { memset(out->buf + out->n, 'I', (size_t)tlen); out->n += tlen; return ST_OK; }
  int rc, n = 0, sc = 0;
  if (score_remaining <= BIALIGN_FALLBACK_MIN_SCORE) {
base:
    if (st) st->base_calls++;
    rc = unialign(p, plen, t, tlen, pen, 0, 0, 0, 0, 0, comp_begin, comp_end, out->buf + out->n, &n, &sc, st);
    if (rc == ST_OK) out->n += n;
    return rc;
  }
  wfo_breakpoint_t bp;
  if (st) st->bialign_calls++;
  rc = find_breakpoint(p, plen, t, tlen, pen, comp_begin, comp_end, &bp, st);
  if (rc == ST_END_REACHED) goto base;
  if (rc != ST_OK) return rc;
  if (bp.score == INT_MAX) return ST_UNREACHABLE;
  const int bp_h = bp.offset_forward, bp_v = bp.offset_forward - bp.k_forward;
  rc = bialign_rec(p, bp_v, t, bp_h, pen, comp_begin, bp.component, bp.score_forward, out, level + 1, st);
  if (rc != ST_OK) return rc;
  return bialign_rec(p + bp_v, plen - bp_v, t + bp_h, tlen - bp_h, pen, bp.component, comp_end,
                     bp.score_reverse, out, level + 1, st);
}

/* ------------------------------------------------------------------ */
/* public API */

int wfo_align_end2end_uni(const char* pattern, int plen, const char* text, int tlen,
                          const wfo_penalties_t* pen, char* ops_out, int* nops, int* score, wfo_stats_t* stats) {
  int rc = unialign(pattern, plen, text, tlen, pen, 0, 0, 0, 0, 0, WFO_M, WFO_M, ops_out, nops, score, stats);
  if (rc == ST_OK) ops_out[*nops] = 0;
  return rc;
}

int wfo_align_end2end_comp(const char* pattern, int plen, const char* text, int tlen,
                           const wfo_penalties_t* pen, int comp_begin, int comp_end,
                           char* ops_out, int* nops, int* score, wfo_stats_t* stats) {
  int rc = unialign(pattern, plen, text, tlen, pen, 0, 0, 0, 0, 0, comp_begin, comp_end, ops_out, nops, score, stats);
  if (rc == ST_OK) ops_out[*nops] = 0;
  return rc;
}

int wfo_align_endsfree(const char* pattern, int plen, int pbf, int pef,
                       const char* text, int tlen, int tbf, int tef,
                       const wfo_penalties_t* pen, char* ops_out, int* nops, int* score, wfo_stats_t* stats) {
  int rc = unialign(pattern, plen, text, tlen, pen, 1, pbf, pef, tbf, tef, WFO_M, WFO_M, ops_out, nops, score, stats);
  if (rc == ST_OK) ops_out[*nops] = 0;
  return rc;
}

int wfo_find_breakpoint(const char* pattern, int plen, const char* text, int tlen,
                        const wfo_penalties_t* pen, int comp_begin, int comp_end,
                        wfo_breakpoint_t* bp, wfo_stats_t* stats) {
  return find_breakpoint(pattern, plen, text, tlen, pen, comp_begin, comp_end, bp, stats);
}

int wfo_find_breakpoint_bounded(const char* pattern, int plen, const char* text, int tlen,
                                const wfo_penalties_t* pen, int comp_begin, int comp_end, int sub,
                                wfo_breakpoint_t* bp, wfo_stats_t* stats) {
  return find_breakpoint_sub(pattern, plen, text, tlen, pen, comp_begin, comp_end, bp, stats, 1, sub, 0, NULL);
}

int wfo_find_breakpoint_rounds(const char* pattern, int plen, const char* text, int tlen,
                               const wfo_penalties_t* pen, int comp_begin, int comp_end, int sub, int tests_per_round,
                               wfo_breakpoint_t* bp, int* rounds, wfo_stats_t* stats) {
  return find_breakpoint_sub(pattern, plen, text, tlen, pen, comp_begin, comp_end, bp, stats, sub >= 0, sub >= 0 ? sub : 0, tests_per_round, rounds);
}

int wfo_align_end2end_biwfa(const char* pattern, int plen, const char* text, int tlen,
                            const wfo_penalties_t* pen, char* ops_out, int* nops, int* score, wfo_stats_t* stats) {
  int rc;
  /* wavefront_bialign(): short sequences go straight to the unidirectional aligner */
  if (MAXI(plen, tlen) <= BIALIGN_FALLBACK_MIN_LENGTH) {
    if (stats) stats->base_calls++;
    rc = unialign(pattern, plen, text, tlen, pen, 0, 0, 0, 0, 0, WFO_M, WFO_M, ops_out, nops, score, stats);
    if (rc == ST_OK) ops_out[*nops] = 0;
    return rc;
  }
  fbuf_t fb; fb.buf = ops_out; fb.n = 0;
  rc = bialign_rec(pattern, plen, text, tlen, pen, WFO_M, WFO_M, INT_MAX, &fb, 0, stats);
  if (rc != ST_OK) return rc;
  *nops = (int)fb.n;
  ops_out[fb.n] = 0;
  *score = (int)wfo_ops_score(ops_out, *nops, pen);
  return ST_OK;
}

int64_t wfo_ops_score(const char* ops, int nops, const wfo_penalties_t* pen) {
  int64_t s = 0;
  int i = 0;
  while (i < nops) {
    char op = ops[i];
    int j = i;
    while (j < nops && ops[j] == op) ++j;
    int64_t L = j - i;
    if (op == 'X') s += L * pen->x;
    else if (op == 'I' || op == 'D') {
      int64_t a = pen->o1 + L * pen->e1, b = pen->o2 + L * pen->e2;
      s += a < b ? a : b;
    }
    i = j;
  }
  return s;
}

int wfo_ops_check(const char* ops, int nops, const char* pattern, int plen, const char* text, int tlen) {
  int v = 0, h = 0;
  for (int i = 0; i < nops; ++i) {
    switch (ops[i]) {
      case 'M': if (v >= plen || h >= tlen || pattern[v] != text[h]) return i + 1; ++v; ++h; break;
      case 'X': if (v >= plen || h >= tlen || pattern[v] == text[h]) return i + 1; ++v; ++h; break;
      case 'I': if (h >= tlen) return i + 1; ++h; break;
      case 'D': if (v >= plen) return i + 1; ++v; break;
      default: return i + 1;
    }
  }
  return (v == plen && h == tlen) ? 0 : -1;
}

/* ------------------------------------------------------------------ */
/* O(nm) DP, 5 states, two rolling rows */
static int64_t dp_generic(const char* p, int plen, const char* t, int tlen, const wfo_penalties_t* pen,
                          int endsfree, int pbf, int pef, int tbf, int tef) {
  const int64_t INF = (int64_t)1 << 50;
  size_t W = (size_t)tlen + 1;
  int64_t* M  = (int64_t*)malloc(2 * W * sizeof(int64_t));
  int64_t* I1 = (int64_t*)malloc(2 * W * sizeof(int64_t));
  int64_t* I2 = (int64_t*)malloc(2 * W * sizeof(int64_t));
  int64_t* D1 = (int64_t*)malloc(2 * W * sizeof(int64_t));
  int64_t* D2 = (int64_t*)malloc(2 * W * sizeof(int64_t));
  int64_t best = INF;
  for (int v = 0; v <= plen; ++v) {
    int64_t *m = M + (v & 1) * W, *i1 = I1 + (v & 1) * W, *i2 = I2 + (v & 1) * W, *d1 = D1 + (v & 1) * W, *d2 = D2 + (v & 1) * W;
    int64_t *pm = M + ((v + 1) & 1) * W, *pd1 = D1 + ((v + 1) & 1) * W, *pd2 = D2 + ((v + 1) & 1) * W;
    for (int h = 0; h <= tlen; ++h) {
      int64_t vi1 = INF, vi2 = INF, vd1 = INF, vd2 = INF, vm = INF;
      if (h > 0) {
        vi1 = MINI(m[h - 1] + pen->o1 + pen->e1, i1[h - 1] + pen->e1);
        vi2 = MINI(m[h - 1] + pen->o2 + pen->e2, i2[h - 1] + pen->e2);
      }
      if (v > 0) {
        vd1 = MINI(pm[h] + pen->o1 + pen->e1, pd1[h] + pen->e1);
        vd2 = MINI(pm[h] + pen->o2 + pen->e2, pd2[h] + pen->e2);
      }
      if (v > 0 && h > 0) vm = pm[h - 1] + (p[v - 1] == t[h - 1] ? 0 : pen->x);
      if (v == 0 && h == 0) vm = 0;
      if (endsfree) {
        if (v == 0 && h <= tbf) vm = 0;
        if (h == 0 && v <= pbf) vm = 0;
      }
      vm = MINI(vm, MINI(MINI(vi1, vi2), MINI(vd1, vd2)));
      if (vi1 > INF) vi1 = INF;
      if (vi2 > INF) vi2 = INF;
      if (vd1 > INF) vd1 = INF;
      if (vd2 > INF) vd2 = INF;
      if (vm > INF) vm = INF;
      m[h] = vm; i1[h] = vi1; i2[h] = vi2; d1[h] = vd1; d2[h] = vd2;
      if (endsfree) {
        if (h == tlen && plen - v <= pef && vm < best) best = vm;
        if (v == plen && tlen - h <= tef && vm < best) best = vm;
      }
    }
  }
  int64_t res = endsfree ? best : M[(plen & 1) * W + tlen];
  free(M); free(I1); free(I2); free(D1); free(D2);
  return res;
}

int64_t wfo_dp_score(const char* pattern, int plen, const char* text, int tlen, const wfo_penalties_t* pen) {
  return dp_generic(pattern, plen, text, tlen, pen, 0, 0, 0, 0, 0);
}

int64_t wfo_dp_score_endsfree(const char* pattern, int plen, const char* text, int tlen,
                              const wfo_penalties_t* pen, int pbf, int pef, int tbf, int tef) {
  return dp_generic(pattern, plen, text, tlen, pen, 1, pbf, pef, tbf, tef);
}

int wfo_align_batch_biwfa(const char* seqs, const int64_t* p_off, const int32_t* p_len,
                          const int64_t* t_off, const int32_t* t_len, int n,
                          const wfo_penalties_t* pen, char* ops_arena, const int64_t* ops_off,
                          int32_t* nops, int32_t* scores, int nthreads, wfo_stats_t* stats_sum) {
  int failed = 0;
  wfo_stats_t tot; memset(&tot, 0, sizeof(tot));
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  #pragma omp parallel for schedule(dynamic, 1) reduction(+ : failed)
  for (int i = 0; i < n; ++i) {
    wfo_stats_t st; memset(&st, 0, sizeof(st));
    int no = 0, sc = 0;
    int rc = wfo_align_end2end_biwfa(seqs + p_off[i], p_len[i], seqs + t_off[i], t_len[i], pen,
                                     ops_arena + ops_off[i], &no, &sc, &st);
    if (rc != 0) { failed++; nops[i] = -1; scores[i] = -1; }
    else { nops[i] = no; scores[i] = sc; }
    #pragma omp critical
    {
      tot.cells += st.cells; tot.extend_bases += st.extend_bases;
      tot.bialign_calls += st.bialign_calls; tot.base_calls += st.base_calls;
      if (st.max_depth > tot.max_depth) tot.max_depth = st.max_depth;
    }
  }
  if (stats_sum) *stats_sum = tot;
  return failed;
}
