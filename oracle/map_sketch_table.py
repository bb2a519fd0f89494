"""TEST INFRASTRUCTURE ONLY -- a model of sketch_fragments_table_kernel's control flow (wfmash_amd/csrc/map_kernels.hip): the
threshold that is raised while too few distinct hashes lie under it, bisected when more lie under it than the table holds,
and the (first position, last position, strand sum) a table slot accumulates.  What the model pins, without a GPU: the loop
terminates for any table size >= the sketch size, and what it returns does not depend on the path the threshold took -- it is
sketchSequence's answer (commonFunc.hpp:218-323; tests/test_map_sketch_table_model.py holds it against the oracle's and,
where oracle/_ref is built, the reference's own).  `cap` is the number of distinct hashes the table accepts (3/4 of its slots
on the device); the tests use tiny tables so that overflows and bisections happen all the time."""
import numpy as np

EMPTY = (1 << 64) - 1
TMAX = EMPTY - 1


def sketch_table(hashes, strands, s, cap, tau0=None):
    """hashes[i], strands[i] (+1 / -1, 0 = no valid k-mer) of the fragment's k-mers -> list of (hash, wpos, wpos_end, strand),
    and the thresholds tried."""
    nk = len(hashes)
    if nk <= 0:
        return [], []
    assert cap >= s, "the table must hold a sketch"
    if tau0 is None:
        t = (2 * s + 24) / (2.0 * nk)
        tau = TMAX if t >= 0.999 else int(t * 18446744073709551616.0)
    else:
        tau = tau0
    lo = hi = 0
    tried = []
    valid = [(int(h), i, int(st)) for i, (h, st) in enumerate(zip(hashes, strands)) if st != 0 and int(h) != EMPTY]
    while True:
        tried.append(tau)
        assert len(tried) <= 200, "the threshold loop does not terminate"
        table = {}
        over = False
        for h, i, st in valid:
            if h > tau:
                continue
            e = table.get(h)
            if e is None:
                if len(table) >= cap:   # the device sets the flag with the insertion that passes the cap and gives up the pass
                    over = True
                    break
                table[h] = [i, i, 1 if st > 0 else -1]
            else:
                e[0] = min(e[0], i); e[1] = max(e[1], i); e[2] += 1 if st > 0 else -1
        if over:
            hi = tau
            tau = lo + (hi - lo) // 2
            continue
        D = len(table)
        if D >= s or tau == TMAX:
            break
        lo = tau
        tau = lo + (hi - lo) // 2 if hi else (TMAX if tau > TMAX // 4 else tau * 4)
    out = []
    for h in sorted(table)[:s]:
        a, b, v = table[h]
        out.append((h, a, b, 1 if v > 0 else (0 if v == 0 else -1)))
    return out, tried
