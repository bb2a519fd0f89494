"""Deterministic synthetic inputs for the BASELINE.json configs.

C3: 64 pairs; target = 50,000 i.i.d. uniform ACGT from splitmix64(seed=0xC3+i);
    query = target mutated at 5 % per-base event rate (80 % substitution,
    10 % insertion, 10 % deletion, indel length ~ Geometric(p=0.5)),
    seed 0xC30000+i.
C5: same generator, 100 kb, 15 % (70/15/15).
C4: 8 haplotypes of one random chr1-sized backbone, per haplotype 0.1 % SNPs, 0.01 % short indels and 20 structural
    variants of 10-100 kb, names hapN#1#chr1 (`pangenome`, vectorised numpy: 2 Gbp have to be made in seconds).
C1: data/scerevisiae8.fa.gz is not in the image; the substitute is `yeast_like`: 8 strains x 16 chromosomes of one
    random genome (12 Mbp at full size), strain-level divergence 0.3-1.2 %, names STRAIN#1#chrN.
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


# ---- C4 / C1: whole-sequence haplotypes (vectorised; seeded numpy Generator) ----

_COMP = np.zeros(256, dtype=np.uint8)
_COMP[_ACGT] = np.frombuffer(b"TGCA", dtype=np.uint8)


def random_backbone(seed: int, n: int) -> np.ndarray:
    return _ACGT[np.random.default_rng(seed).integers(0, 4, n, dtype=np.uint8)]


def haplotype(base: np.ndarray, seed: int, snp=1e-3, indel=1e-4, n_sv=20, sv_min=10_000, sv_max=100_000) -> np.ndarray:
    """One haplotype of `base` (uint8 array over ACGT): SNPs at rate `snp`, short indels (1 + Geometric(0.5) bases, half
    deletions, half insertions) at rate `indel`, then n_sv structural variants of sv_min..sv_max bases (deletion /
    tandem duplication / inversion)."""
    rng = np.random.default_rng(seed)
    s = base.copy()
    n = len(s)
    pos = rng.integers(0, n, int(n * snp))
    code = np.searchsorted(_ACGT, s[pos])  # ACGT is sorted
    s[pos] = _ACGT[(code + rng.integers(1, 4, len(pos))) % 4]
    pos = np.unique(rng.integers(0, n, int(n * indel)))
    ln = rng.geometric(0.5, len(pos))
    is_del = rng.random(len(pos)) < 0.5
    keep = np.ones(n, dtype=bool)
    for p, l in zip(pos[is_del], ln[is_del]):
        keep[p:p + l] = False
    ins_pos = np.repeat(pos[~is_del], ln[~is_del])
    ins_val = _ACGT[rng.integers(0, 4, len(ins_pos))]
    # np.insert indexes into the ORIGINAL array; deleted bases are removed afterwards through `keep`
    keep = np.insert(keep, ins_pos, True)
    s = np.insert(s, ins_pos, ins_val)[keep]
    for _ in range(n_sv):
        l = int(rng.integers(sv_min, sv_max + 1))
        if l >= len(s) // 2:
            continue
        p = int(rng.integers(0, len(s) - l))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            s = np.concatenate([s[:p], s[p + l:]])
        elif kind == 1:
            s = np.concatenate([s[:p + l], s[p:p + l], s[p + l:]])
        else:
            s[p:p + l] = _COMP[s[p:p + l]][::-1]
    return s


def pangenome(n_haps=8, length=248_956_422, seed=0xC4, **kw):
    """C4: yields (name, uint8 array) for hap1#1#chr1 .. hapN#1#chr1 (haplotype seeds 0xC400 + h)."""
    base = random_backbone(seed, length)
    for h in range(n_haps):
        yield f"hap{h + 1}#1#chr1", haplotype(base, (seed << 8) + h, **kw)


def pangenome_parallel(n_haps=8, length=248_956_422, seed=0xC4, workers=8, **kw):
    """pangenome() with the haplotypes made side by side (numpy releases the GIL in the passes that matter): the same records,
    as a list.  A chr1-sized set is 8 x 8 s of numpy one after the other."""
    from concurrent.futures import ThreadPoolExecutor
    base = random_backbone(seed, length)
    with ThreadPoolExecutor(max(1, min(workers, n_haps))) as ex:
        haps = list(ex.map(lambda h: haplotype(base, (seed << 8) + h, **kw), range(n_haps)))
    return [(f"hap{h + 1}#1#chr1", haps[h]) for h in range(n_haps)]


YEAST_STRAINS = ["S288C", "DBVPG6044", "DBVPG6765", "SK1", "UWOPS034614", "Y12", "YPS128", "Y55"]  # 8 names as in scerevisiae8


def yeast_like(n_strains=8, n_chrom=16, genome_bp=12_000_000, seed=0xC1):
    """C1 substitute: yields (name, uint8 array) for STRAIN#1#chrK; chromosome lengths spread 1 : 6 like yeast's, the first
    strain is the backbone itself, the others diverge from it by 0.3-1.2 % SNPs, a tenth of that in short indels, and
    two structural variants per chromosome."""
    w = np.linspace(1.0, 6.0, n_chrom)
    lens = np.maximum(2000, (w / w.sum() * genome_bp).astype(np.int64))
    chroms = [random_backbone(seed * 1000 + c, int(lens[c])) for c in range(n_chrom)]
    for st in range(n_strains):
        rate = 0.003 + 0.009 * st / max(1, n_strains - 1)
        for c in range(n_chrom):
            sv = int(min(20_000, max(500, lens[c] // 40)))
            seq = chroms[c] if st == 0 else haplotype(chroms[c], seed * 100000 + st * 100 + c, snp=rate, indel=rate / 10, n_sv=2,
                                                      sv_min=sv // 4, sv_max=sv)
            yield f"{YEAST_STRAINS[st % len(YEAST_STRAINS)]}#1#chr{c + 1}", seq


def write_fasta(path, records, fai=True):
    """records: iterable of (name, bytes or uint8 array); one line per sequence + .fai.  Returns (names, lengths)."""
    names, lengths = [], []
    with open(path, "wb") as f, open(path + ".fai", "w") as idx:
        for name, s in records:
            s = np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else s
            hdr = f">{name}\n".encode()
            off = f.tell() + len(hdr)
            f.write(hdr)
            s.tofile(f)
            f.write(b"\n")
            idx.write(f"{name}\t{len(s)}\t{off}\t{len(s)}\t{len(s) + 1}\n")
            names.append(name)
            lengths.append(len(s))
    if not fai:
        import os
        os.unlink(path + ".fai")
    return names, lengths
