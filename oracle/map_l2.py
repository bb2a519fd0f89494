"""TEST INFRASTRUCTURE ONLY -- restatement of the L2 stage of mashmap3 as wfmash runs it:
  SlideMapper                 src/map/include/slidingMap.hpp:28-212
  computeL2MappedRegions      src/map/include/mappingCore.hpp:307-442
  doL2Mapping                 src/map/include/computeMap.hpp:989-1061
  mapSingleQueryFrag (L2 part) src/map/include/computeMap.hpp:895-921
PARITY UNPINNED: these headers cannot be compiled here (winSketch.hpp pulls htslib), so there
is no golden vector from the reference itself; the restatement follows the source line by line,
including libstdc++'s push_heap/pop_heap element order (the order in which expired reference
minmers leave the window decides how the lazy pivot moves)."""
import numpy as np

from oracle import map_stats as MS

FWD, REV = 1, -1


# --- libstdc++ heap algorithms (bits/stl_heap.h: __push_heap, __adjust_heap), comp(l, r) = l.wpos_end > r.wpos_end
def _push_heap_at(a, hole, top, value, comp):
    parent = (hole - 1) // 2
    while hole > top and comp(a[parent], value):
        a[hole] = a[parent]
        hole = parent
        parent = (hole - 1) // 2
    a[hole] = value


def push_heap(a, comp):
    _push_heap_at(a, len(a) - 1, 0, a[-1], comp)


def pop_heap(a, comp):
    """moves the top to a[-1]; caller pops it."""
    if len(a) > 1:
        value = a[-1]
        a[-1] = a[0]
        n = len(a) - 1
        hole = 0
        child = 0
        while child < (n - 1) // 2:
            child = 2 * (child + 1)
            if comp(a[child], a[child - 1]):
                child -= 1
            a[hole] = a[child]
            hole = child
        if (n & 1) == 0 and child == (n - 2) // 2:
            child = 2 * (child + 1)
            a[hole] = a[child - 1]
            hole = child - 1
        _push_heap_at(a, hole, 0, value, comp)


class SlideMapper:
    """slidingMap.hpp:28-212.  q_minmers: list of (hash, strand) ascending by hash."""

    def __init__(self, q_minmers):
        self.S = len(q_minmers)
        # slot 0 is the value-initialised sentinel
        self.hash = [0] + [m[0] for m in q_minmers]
        self.q_strand = [0] + [m[1] for m in q_minmers]
        self.vote = [0] * (self.S + 1)
        self.nbi = [0] + [1] * self.S
        self.active = [0] * (self.S + 1)
        self.pivot = self.S
        self.piv_rank = self.S
        self.shared = 0
        self.strand_votes = 0
        self.isect = 0

    def _loc(self, h):
        lo, hi = 1, self.S + 1
        while lo < hi:
            mid = (lo + hi) // 2
            if self.hash[mid] < h:
                lo = mid + 1
            else:
                hi = mid
        return lo

    def insert(self, h, strand):
        i = self._loc(h)
        if i == self.S + 1:
            return
        if self.hash[i] == h:
            self.active[i] = 1
            self.vote[i] += self.q_strand[i] * strand
            self.isect += 1
            if self.hash[i] <= self.hash[self.pivot]:
                self.shared += 1
                self.strand_votes += self.vote[i]
        else:
            self.nbi[i] += 1
            if self.hash[i] <= self.hash[self.pivot]:
                self.piv_rank += 1
            if self.piv_rank > self.S:
                self.shared -= self.active[self.pivot]
                self.strand_votes -= self.vote[self.pivot]
                self.piv_rank -= self.nbi[self.pivot]
                self.pivot -= 1

    def delete(self, h, strand):
        i = self._loc(h)
        if i == self.S + 1:
            return
        if self.hash[i] == h:
            if self.hash[i] <= self.hash[self.pivot]:
                self.shared -= 1
                self.strand_votes -= self.vote[i]
            self.active[i] = 0
            self.vote[i] = 0
            self.isect -= 1
        else:
            self.nbi[i] -= 1
            if self.hash[i] <= self.hash[self.pivot]:
                self.piv_rank -= 1
            if self.pivot + 1 != self.S + 1 and self.piv_rank + self.nbi[self.pivot + 1] <= self.S:
                self.pivot += 1
                self.shared += self.active[self.pivot]
                self.strand_votes += self.vote[self.pivot]
                self.piv_rank += self.nbi[self.pivot]


def _lower_bound(index, seq_id, wpos):
    lo, hi = 0, len(index)
    while lo < hi:
        mid = (lo + hi) // 2
        if (index[mid][3], index[mid][1]) < (seq_id, wpos):
            lo = mid + 1
        else:
            hi = mid
    return lo


def l2_mapped_regions(q_minmers, q_len, cand, index, window_length):
    """computeL2MappedRegions.  index: minmerIndex as a list of (hash, wpos, wpos_end, seqId, strand);
    cand: dict(seqId, start, end, isect).  Returns a list of L2 loci (dicts).  window length 0 only."""
    assert q_len == window_length
    w = window_length
    it = _lower_bound(index, cand["seqId"], cand["start"] - w - 1)
    n = len(index)
    heap = []
    comp = lambda l, r: l[2] > r[2]
    sm = SlideMapper(q_minmers)
    best_sketch = 1
    in_cand = False
    l2 = dict(seqId=0, mean=0, start=0, end=0, shared=0, strand=0)
    out = []

    def finish(l2, strand_votes, seq_id):
        l2["mean"] = (l2["start"] + l2["end"]) // 2
        l2["seqId"] = seq_id
        l2["strand"] = FWD if strand_votes >= 0 else REV
        if not out or out[-1]["end"] + w < l2["start"]:
            out.append(dict(l2))
        else:
            out[-1]["end"] = l2["end"]
            out[-1]["mean"] = (out[-1]["start"] + out[-1]["end"]) // 2

    while it != n and index[it][3] == cand["seqId"] and index[it][1] < cand["start"]:
        if index[it][2] > cand["start"]:
            heap.append(index[it])
            push_heap(heap, comp)
            sm.insert(index[it][0], index[it][4])
        it += 1
    while it != n and index[it][3] == cand["seqId"] and index[it][1] <= cand["end"]:
        prev_votes = sm.strand_votes
        while heap and heap[0][2] <= index[it][1]:
            sm.delete(heap[0][0], heap[0][4])
            pop_heap(heap, comp)
            heap.pop()
        sm.insert(index[it][0], index[it][4])
        heap.append(index[it])
        push_heap(heap, comp)
        if sm.shared > best_sketch:
            out.clear()
            in_cand = True
            best_sketch = sm.shared
            l2["shared"] = sm.shared
            l2["start"] = index[it][1]
            l2["end"] = index[it][1]
        elif sm.shared == best_sketch:
            if not in_cand:
                l2["shared"] = sm.shared
                l2["start"] = index[it][1]
            in_cand = True
            l2["end"] = index[it][1]
        else:
            if in_cand:
                finish(l2, prev_votes, index[it][3])
                l2 = dict(seqId=0, mean=0, start=0, end=0, shared=0, strand=0)
            in_cand = False
        it += 1
    if in_cand:
        finish(l2, sm.strand_votes, index[it - 1][3])
    return out


def cutoff_j(q_sketch_size, k, ani_diff=0.0, hg_numerator=1.0):
    """computeMap.hpp:1001-1004 (double arithmetic around the float j2md/md2j)."""
    jac = hg_numerator / q_sketch_size
    mash = float(MS.j2md(np.float32(jac), k))
    cutoff_ani = max(0.0, (1 - mash) - float(np.float32(ani_diff)))
    return float(MS.md2j(np.float32(1 - cutoff_ani), k))


def identity_tables(sketch_size, k, pct_identity, keep_low_pct_id=True, ci=0.95):
    """(keep, nucIdentity x 1e4) for every (Q.sketchSize, sharedSketchSize) (computeMap.hpp:1018-1036)."""
    f32 = np.float32
    S = sketch_size
    keep = np.zeros((S + 1, S + 1), dtype=np.uint8)
    ident = np.zeros((S + 1, S + 1), dtype=np.uint16)
    pi = f32(pct_identity)
    for qs in range(1, S + 1):
        for sh in range(0, qs + 1):
            mash = MS.j2md(f32(1.0 * sh / qs), k)
            nuc = f32(f32(1) - mash)
            ub = f32(f32(1) - MS.md_lower_bound(mash, qs, k, ci))
            keep[qs, sh] = 1 if ((keep_low_pct_id and ub >= pi) or nuc >= pi) else 0
            v = f32(nuc * f32(10000.0))
            # static_cast<uint16_t>(roundf(...)): roundf = half away from zero
            ident[qs, sh] = int(np.floor(abs(float(v)) + 0.5)) if v >= 0 else 0
    return keep, ident


def do_l2_mapping(q_minmers, q_len, kc_u8, cands, index, ref_group, params):
    """mapSingleQueryFrag's L2 half for one fragment: candidates grouped as in computeMap.hpp:895-918,
    best-first with the ANI cutoff, identity filter, final sort by (refSeqId, refStartPos).
    Returns tuples (refSeqId, refStartPos, queryStartPos, blockLength, n_merged, conservedSketches,
    nucIdentity, flags, kmerComplexity) = skch::MappingResult (base_types.hpp:154-165)."""
    qs = len(q_minmers)
    res = []
    keep, ident = params["keep_table"], params["ident_table"]
    b = 0
    while b < len(cands):
        if params["skip_prefix"]:
            g = ref_group[cands[b]["seqId"]]
            e = b
            while e < len(cands) and ref_group[cands[e]["seqId"]] == g:
                e += 1
        else:
            e = len(cands)
        run = cands[b:e]
        if params["stage1_topani"]:
            run = sorted(run, key=lambda c: -c["isect"])  # best-first; stops at the first one below the cutoff
        for c in run:
            if params["stage1_topani"] and c["isect"] / qs < params["cutoff_j"][qs]:
                break
            for l2 in l2_mapped_regions(q_minmers, q_len, c, index, params["window_length"]):
                if keep[qs, l2["shared"]]:
                    res.append((l2["seqId"], l2["mean"] & 0xFFFFFFFF, 0, q_len, 1, l2["shared"], int(ident[qs, l2["shared"]]),
                                1 if l2["strand"] == REV else 0, kc_u8))
        b = e
    res.sort(key=lambda r: (r[0], r[1]))
    return res


def kmer_complexity(last_hash, n_minmers, q_len, k):
    """getSeedHits (mappingCore.hpp:72-74): float Q.kmerComplexity and its uint8 x100 form."""
    max_hash_01 = float(np.longdouble(last_hash) / np.longdouble(0xFFFFFFFFFFFFFFFF))
    kc = np.float32((float(n_minmers) / max_hash_01) / ((q_len - k + 1) * 2))
    v = np.float32(kc * np.float32(100.0))
    return kc, int(np.floor(float(v) + 0.5)) & 0xFF
