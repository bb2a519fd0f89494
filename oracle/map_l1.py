"""TEST INFRASTRUCTURE ONLY -- restatement of the L1 stage of mashmap3 as wfmash runs it:
  getSeedIntervalPoints       src/map/include/mappingCore.hpp:82-131
  computeL1CandidateRegions   src/map/include/mappingCore.hpp:137-301
  doL1Mapping (group loop)    src/map/include/computeMap.hpp:945-984
PARITY UNPINNED: mappingCore.hpp cannot be compiled here (it pulls htslib through
winSketch.hpp/seqiter.hpp), so there is no golden vector from the reference itself for this
stage; the restatement follows the source line by line (pure-Python loops, small cases)."""
import heapq

OPEN, CLOSE = 1, -1
SS_TABLE_MAX = 1000.0


def seed_interval_points(q_minmers, lookup, q_seq_id, ref_group, skip_self=True, skip_prefix=True, lower_triangular=False):
    """k-way merge of the query hashes' point lists by (seqId, pos, side) with the group filters."""
    lists = [lookup[h] for h in q_minmers if h in lookup]
    merged = list(heapq.merge(*lists, key=lambda p: (p[2], p[0], p[3])))
    out = []
    qg = ref_group[q_seq_id]
    for p in merged:
        tg = ref_group[p[2]]
        skip = (skip_self and qg == tg) or (skip_prefix and qg == tg) or (lower_triangular and q_seq_id <= p[2])
        if not skip:
            out.append(p)
    return out


def l1_candidates(points, q_len, q_sketch_size, minimum_hits, window_length, sketch_size, sketch_cutoffs,
                  stage1_topani=True, stage2_full_scan=True, l1=None):
    """computeL1CandidateRegions on one group's points (list of [pos, hash, seqId, side])."""
    if l1 is None:
        l1 = []
    n = len(points)
    if n == 0:
        return l1
    window_len = max(0, q_len - window_length)
    cluster_len = window_length
    freq = {}

    def trailing_ok(t, l):
        return (points[t][2] == points[l][2] and points[t][0] <= points[l][0] - window_len) or points[t][2] < points[l][2]

    overlap = 0
    best = 0
    if stage1_topani:
        t = l = 0
        while l != n:
            while t != n and trailing_ok(t, l):
                if points[t][3] == CLOSE:
                    if window_len != 0:
                        freq[points[t][1]] = freq.get(points[t][1], 0) - 1
                    if window_len == 0 or freq.get(points[t][1], 0) == 0:
                        overlap -= 1
                t += 1
            cur = points[l][0]
            while l != n and points[l][0] == cur:
                if points[l][3] == OPEN:
                    if window_len == 0 or freq.get(points[l][1], 0) == 0:
                        overlap += 1
                    if window_len != 0:
                        freq[points[l][1]] = freq.get(points[l][1], 0) + 1
                l += 1
            best = max(best, overlap)
        if best < minimum_hits:
            return l1
        idx = int(min(best, q_sketch_size) / max(1.0, sketch_size / SS_TABLE_MAX))
        minimum_hits = max(sketch_cutoffs[idx], minimum_hits)
    freq = {}
    best = min(best, q_sketch_size)
    in_cand = False
    cand = dict(seqId=0, start=0, end=0, isect=0)
    local = []
    t = l = 0
    overlap = 0
    prev_overlap = 0
    prev_pos = (0, 0)
    cur_pos = (points[0][2], points[0][0])
    while l != n:
        prev_overlap = overlap
        while t != n and trailing_ok(t, l):
            if points[t][3] == CLOSE:
                if window_len != 0:
                    freq[points[t][1]] = freq.get(points[t][1], 0) - 1
                if window_len == 0 or freq.get(points[t][1], 0) == 0:
                    overlap -= 1
            t += 1
        if points[l][0] != cur_pos[1]:
            prev_pos = cur_pos
            cur_pos = (points[l][2], points[l][0])
        while l != n and points[l][0] == cur_pos[1]:
            if points[l][3] == OPEN:
                if window_len == 0 or freq.get(points[l][1], 0) == 0:
                    overlap += 1
                if window_len != 0:
                    freq[points[l][1]] = freq.get(points[l][1], 0) + 1
            l += 1
        if prev_overlap >= minimum_hits:
            if cand["seqId"] != prev_pos[0] and in_cand:
                local.append(dict(cand))
                cand = dict(seqId=0, start=0, end=0, isect=0)
                in_cand = False
            if not in_cand:
                cand = dict(seqId=prev_pos[0], start=prev_pos[1] - window_len, end=prev_pos[1] - window_len, isect=prev_overlap)
                in_cand = True
            else:
                if stage2_full_scan:
                    cand["isect"] = max(cand["isect"], prev_overlap)
                    cand["end"] = prev_pos[1] - window_len
                elif cand["isect"] < prev_overlap:
                    cand["isect"] = prev_overlap
                    cand["start"] = prev_pos[1] - window_len
                    cand["end"] = prev_pos[1] - window_len
        else:
            if in_cand:
                local.append(dict(cand))
                cand = dict(seqId=0, start=0, end=0, isect=0)
            in_cand = False
    if in_cand:
        local.append(dict(cand))
    for c in local:
        if not l1 or c["seqId"] != l1[-1]["seqId"] or c["start"] > l1[-1]["end"] + cluster_len:
            l1.append(dict(c))
        else:
            l1[-1]["end"] = c["end"]
            l1[-1]["isect"] = max(c["isect"], l1[-1]["isect"])
    return l1


def do_l1_mapping(q_minmers, q_len, q_seq_id, lookup, ref_group, params):
    """doL1Mapping for one fragment.  params: dict(window_length, sketch_size, min_hits_cached,
    cached_segment_length, min_hits_by_qsketch, sketch_cutoffs, skip_self, skip_prefix,
    lower_triangular, stage1_topani, stage2_full_scan)."""
    qs = len(q_minmers)
    if qs == 0:
        return []
    pts = seed_interval_points(q_minmers, lookup, q_seq_id, ref_group, params["skip_self"], params["skip_prefix"], params["lower_triangular"])
    min_hits = params["min_hits_cached"] if q_len == params["cached_segment_length"] else params["min_hits_by_qsketch"][qs]
    l1 = []
    b = 0
    while b < len(pts):
        if params["skip_prefix"]:
            g = ref_group[pts[b][2]]
            e = b
            while e < len(pts) and ref_group[pts[e][2]] == g:
                e += 1
        else:
            e = len(pts)
        l1_candidates(pts[b:e], q_len, qs, min_hits, params["window_length"], params["sketch_size"], params["sketch_cutoffs"],
                      params["stage1_topani"], params["stage2_full_scan"], l1)
        b = e
    return l1
