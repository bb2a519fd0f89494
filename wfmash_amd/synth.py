"""Deterministic synthetic inputs for the BASELINE.json configs.

C3: 64 pairs; target = 50,000 i.i.d. uniform ACGT from splitmix64(seed=0xC3+i);
    query = target mutated at 5 % per-base event rate (80 % substitution,
    10 % insertion, 10 % deletion, indel length ~ Geometric(p=0.5)),
    seed 0xC30000+i.
C5: same generator, 100 kb, 15 % (70/15/15).
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_dna(seed: int, n: int) -> bytes:
    return _ACGT[(splitmix64(seed, n) >> np.uint64(62)).astype(np.int64)].tobytes()


def mutate(seq: bytes, rate: float, seed: int, p_sub=0.8, p_ins=0.1) -> bytes:
    """Sequential mutation process, deterministic in (seq, rate, seed)."""
    n = len(seq)
    r = splitmix64(seed, 4 * n + 16)
    u = (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    ev, typ, lenu, basu = u[0:n], u[n:2 * n], u[2 * n:3 * n], r[3 * n:4 * n]
    pos = np.nonzero(ev < rate)[0]
    src = np.frombuffer(seq, dtype=np.uint8)
    out = []
    cur = 0
    extra_i = 0
    extra = splitmix64(seed ^ 0x5DEECE66D, 64 + 4 * len(pos))
    for p in pos:
        p = int(p)
        if p < cur:
            continue  # inside a deleted stretch
        out.append(src[cur:p])
        t = typ[p]
        L = 1 + int(np.floor(np.log(max(lenu[p], 1e-300)) / np.log(0.5)))
        if t < p_sub:
            b = src[p]
            alt = _ACGT[_ACGT != b]
            out.append(alt[int(basu[p] % np.uint64(3))][None])
            cur = p + 1
        elif t < p_sub + p_ins:
            ins = _ACGT[(extra[extra_i:extra_i + L] >> np.uint64(62)).astype(np.int64)]
            extra_i = (extra_i + L) % (len(extra) - 64)
            out.append(ins)
            cur = p
            # keep the base itself
            out.append(src[p:p + 1])
            cur = p + 1
        else:
            cur = min(n, p + L)
    out.append(src[cur:])
    return np.concatenate(out).tobytes() if out else b""


def pairs(config: str, n_pairs=None, length=None, rate=None):
    """Returns [(target, query)] for 'C3' or 'C5' (pattern = target, text = query)."""
    if config == "C3":
        base, L, r, ps, pi = 0xC3, 50000, 0.05, 0.8, 0.1
        n = 64
    elif config == "C5":
        base, L, r, ps, pi = 0xC5, 100000, 0.15, 0.7, 0.15
        n = 64
    else:
        raise ValueError(config)
    n = n_pairs if n_pairs is not None else n
    L = length if length is not None else L
    r = rate if rate is not None else r
    out = []
    for i in range(n):
        t = random_dna(base + i, L)
        q = mutate(t, r, (base << 16) + i, ps, pi)
        out.append((t, q))
    return out
