"""TEST INFRASTRUCTURE ONLY -- a model of l1_sweep_wave_kernel's control flow (wfmash_amd/csrc/map_l1.hip: l1_chunk,
l1_group_wave), lane by lane in plain Python, so that its claim can be checked without a GPU: the overlap count after a
position group of computeL1CandidateRegions (mappingCore.hpp:137-301) is

    (OPEN points up to the group's end) - (CLOSE points up to the end of the group's first (seq, pos) run)

-- the reference's trailing pointer compares with the group's FIRST point (:214-221), groups are runs of equal pos
whatever the seq (:223-226) -- and the candidate bookkeeping only needs, in order, the groups whose count reaches
minimum_hits plus the fact that something lower lies between two of them.  `width` is the wave's width (64 on the device;
the tests also use 3 .. 8 so that groups, runs and candidates straddle chunk borders all the time).
tests/test_map_l1_wave_model.py holds it against oracle/map_l1.py::l1_candidates."""

OPEN, CLOSE = 1, -1
SS_TABLE_MAX = 1000.0


def _incl_sum(v):
    out, run = [], 0
    for x in v:
        run += x
        out.append(run)
    return out


def _incl_max(v):
    out, run = [], None
    for x in v:
        run = x if run is None else max(run, x)
        out.append(run)
    return out


class _Carry:
    def __init__(self):
        self.opens = 0
        self.sends = 0
        self.gs_sends = 0
        self.gs_seq = 0
        self.closes_passed = 0
        self.last = None  # the key before the chunk


def _chunk(points, base, n, width, cy):
    """l1_chunk: per lane (overlap after the group, pos, seq of the group's first key, ends a group)."""
    lanes = range(width)
    gi = [base + l for l in lanes]
    valid = [g < n for g in gi]
    key = [points[g] if v else None for g, v in zip(gi, valid)]
    nxt0 = points[base + width] if base + width < n else None
    kp = [cy.last] + key[:-1]
    kn = key[1:] + [nxt0]
    pos = [k[0] if k else 0 for k in key]
    seq = [k[2] if k else 0 for k in key]
    opn = [1 if (v and k[3] == OPEN) else 0 for k, v in zip(key, valid)]
    gstart = [v and (g == 0 or kp[l][0] != pos[l]) for l, (g, v) in enumerate(zip(gi, valid))]
    last = [g == n - 1 for g in gi]
    gend = [v and (last[l] or kn[l][0] != pos[l]) for l, v in enumerate(valid)]
    send = [v and (last[l] or kn[l][0] != pos[l] or kn[l][2] != seq[l]) for l, v in enumerate(valid)]
    O = [cy.opens + x for x in _incl_sum(opn)]
    Cc = [g + 1 - o for g, o in zip(gi, O)]
    s_incl = _incl_sum([1 if x else 0 for x in send])
    S_excl = [cy.sends + a - (1 if b else 0) for a, b in zip(s_incl, send)]
    gsl = _incl_max([l if gstart[l] else -1 for l in lanes])
    Sgs = [S_excl[g] if g >= 0 else cy.gs_sends for g in gsl]
    seq_first = [seq[g] if g >= 0 else cy.gs_seq for g in gsl]
    first_send = [send[l] and S_excl[l] == Sgs[l] for l in lanes]
    passed = [max(cy.closes_passed, x) for x in _incl_max([Cc[l] if first_send[l] else -1 for l in lanes])]
    ov = [O[l] - passed[l] for l in lanes]
    w = width - 1
    cy.opens = O[w]
    cy.sends = S_excl[w] + (1 if send[w] else 0)
    cy.gs_sends = Sgs[w]
    cy.gs_seq = seq_first[w]
    cy.closes_passed = passed[w]
    cy.last = key[min(width, n - base) - 1]
    return ov, pos, seq_first, gend


def l1_candidates_wave(points, q_sketch_size, minimum_hits, window_length, sketch_size, sketch_cutoffs, stage1_topani=True,
                       stage2_full_scan=True, l1=None, width=64):
    """l1_group_wave on one group's points ([pos, hash, seqId, side], sorted by (seqId, pos, side)); fragments are window_length
    long (the only case the product accepts), so the reference's window is empty."""
    if l1 is None:
        l1 = []
    n = len(points)
    if n == 0:
        return l1
    if stage1_topani:
        best, cy = 0, _Carry()
        for base in range(0, n, width):
            ov, _, _, gend = _chunk(points, base, n, width, cy)
            best = max([best] + [o for o, g in zip(ov, gend) if g])
        if best < minimum_hits:
            return l1
        idx = int(min(best, q_sketch_size) / max(1.0, sketch_size / SS_TABLE_MAX))
        minimum_hits = max(sketch_cutoffs[min(idx, len(sketch_cutoffs) - 1)], minimum_hits)
    state = dict(in_cand=False, c=dict(seqId=0, start=0, end=0, isect=0))

    def flush(c):
        if not l1 or c["seqId"] != l1[-1]["seqId"] or c["start"] > l1[-1]["end"] + window_length:
            l1.append(dict(c))
        else:
            l1[-1]["end"] = c["end"]
            l1[-1]["isect"] = max(c["isect"], l1[-1]["isect"])

    cy = _Carry()
    for base in range(0, n, width):
        ov, pos, sf, gend = _chunk(points, base, n, width, cy)
        elem = [gend[l] and base + l != n - 1 for l in range(width)]
        E = [l for l in range(width) if elem[l]]
        H = [l for l in E if ov[l] >= minimum_hits]
        if not H:
            if state["in_cand"] and E:
                flush(state["c"])
                state["c"] = dict(seqId=0, start=0, end=0, isect=0)
                state["in_cand"] = False
            continue
        while E:
            if not state["in_cand"]:  # straight to the next element that reaches minimum_hits
                rest = [l for l in E if l in H]
                if not rest:
                    break
                E = [l for l in E if l >= rest[0]]
            e = E.pop(0)
            o = ov[e]
            c = state["c"]
            if o >= minimum_hits:
                pp, sq = pos[e], sf[e]
                if c["seqId"] != sq and state["in_cand"]:
                    flush(c)
                    c = state["c"] = dict(seqId=0, start=0, end=0, isect=0)
                    state["in_cand"] = False
                if not state["in_cand"]:
                    c.update(seqId=sq, start=pp, end=pp, isect=o)
                    state["in_cand"] = True
                elif stage2_full_scan:
                    c["isect"] = max(c["isect"], o)
                    c["end"] = pp
                elif c["isect"] < o:
                    c.update(isect=o, start=pp, end=pp)
            else:
                if state["in_cand"]:
                    flush(c)
                    state["c"] = dict(seqId=0, start=0, end=0, isect=0)
                state["in_cand"] = False
    if state["in_cand"]:
        flush(state["c"])
    return l1
