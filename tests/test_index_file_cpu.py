"""The on-disk index of the map phase (`-W` / `-I`; wfmash_amd/host/index_file.cpp) without a GPU:
the id section against the reference's own exportIdMapping (golden bytes, and live where oracle/_ref is built),
and reader + writer against the restatement of the reference's read/write code (oracle/map_index_file.py)."""
import os
import random

import numpy as np
import pytest

from oracle import map_index_file as IF
from oracle import pyfilter
from wfmash_amd import capi

pytestmark = pytest.mark.skipif(not os.path.exists(capi.LIB_PATH), reason="libwfmash_hip.so not built")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "index_ids_golden.bin")


def write_names(dirname, n=137, seed=11):
    """an (empty) FASTA with a .fai of n PanSN-style names: enough for the name -> id map to rehash several times"""
    rng = random.Random(seed)
    fa = os.path.join(dirname, "names.fa")
    open(fa, "w").close()
    with open(fa + ".fai", "w") as f:
        for i in range(n):
            f.write(f"sample{rng.randrange(40)}#{1 + i % 2}#ctg{i:04d}_{rng.randrange(10**6)}\t{1000 + 37 * i}\t0\t60\t61\n")
    return fa


def test_id_section_equals_the_references(tmp_path):
    fa = write_names(str(tmp_path))
    mine = str(tmp_path / "mine.bin")
    capi.host_index_file("ids", fa, mine)
    got = open(mine, "rb").read()
    assert got == open(GOLDEN, "rb").read()  # tests/golden/make_index_ids_golden.py
    entries, next_id, end = IF.parse_ids(got)
    assert end == len(got) and next_id == 137 and sorted(i for _, i in entries) == list(range(137))
    if pyfilter.have_ref():
        ref = str(tmp_path / "ref.bin")
        pyfilter.ref_export_ids(fa, ref)
        assert open(ref, "rb").read() == got


def random_sub(rng, names, ids_bytes, bidx, btotal, n_minmers, w, s, k):
    m = np.zeros(n_minmers, dtype=IF.MINMER)
    hashes = rng.integers(1, 2**63, max(1, n_minmers // 3), dtype=np.int64).astype(np.uint64)
    m["hash"] = rng.choice(hashes, n_minmers)
    m["seqId"] = np.sort(rng.integers(0, len(names), n_minmers))
    m["wpos"] = rng.integers(0, 100000, n_minmers)
    m["wpos_end"] = m["wpos"] + rng.integers(1, w, n_minmers)
    m["strand"] = rng.choice([-1, 1], n_minmers)
    keys = [int(x) for x in dict.fromkeys(m["hash"].tolist())]
    lists = []
    for key in keys:
        rows = m[m["hash"] == key]
        p = np.zeros(2 * len(rows), dtype=IF.POINT)
        p["pos"][0::2] = rows["wpos"]; p["pos"][1::2] = rows["wpos_end"]
        p["hash"] = key
        p["seqId"][0::2] = rows["seqId"]; p["seqId"][1::2] = rows["seqId"]
        p["side"][0::2] = 1; p["side"][1::2] = -1
        lists.append(p)
    order = list(range(len(keys)))
    random.Random(int(rng.integers(1 << 30))).shuffle(order)  # the reference's key order depends on its thread count
    return dict(batch_idx=bidx, total_batches=btotal, batch_size=5_000_000, names=names, ids_bytes=ids_bytes, w=w, s=s, k=k, minmers=m,
                keys=[keys[i] for i in order], lists=[lists[i] for i in order])


def test_reader_and_writer_against_the_restated_format(tmp_path):
    fa = write_names(str(tmp_path), n=9)
    ids_path = str(tmp_path / "ids.bin")
    capi.host_index_file("ids", fa, ids_path)
    ids_bytes = open(ids_path, "rb").read()
    names = [n for n, _ in sorted(IF.parse_ids(ids_bytes)[0], key=lambda e: e[1])]
    rng = np.random.default_rng(3)
    subs = [random_sub(rng, names[:5], ids_bytes, 0, 2, 4000, 1000, 39, 15), random_sub(rng, names[5:], ids_bytes, 1, 2, 1500, 1000, 39, 15)]
    src, out, out2 = (str(tmp_path / n) for n in ("in.idx", "out.idx", "out2.idx"))
    IF.write(src, subs)
    capi.host_index_file("rewrite", fa, out, in_path=src)
    got = IF.parse(out)
    assert len(got) == 2
    for a, b in zip(got, subs):
        for f in ("batch_idx", "total_batches", "batch_size", "names", "ids_bytes", "w", "s", "k"):
            assert a[f] == b[f], f
        assert a["minmers"].tobytes() == b["minmers"].tobytes()
        # keys: each once, in the order of their first interval in minmerIndex; point lists unchanged
        first = {}
        for i, h in enumerate(b["minmers"]["hash"].tolist()):
            first.setdefault(h, i)
        assert a["keys"] == sorted(b["keys"], key=lambda h: first[h])
        want = dict(zip(b["keys"], b["lists"]))
        for key, pts in zip(a["keys"], a["lists"]):
            assert pts.tobytes() == want[key].tobytes()
    capi.host_index_file("rewrite", fa, out2, in_path=out)
    assert open(out2, "rb").read() == open(out, "rb").read()


def test_reader_rejects_damaged_files(tmp_path):
    fa = write_names(str(tmp_path), n=4)
    ids_path = str(tmp_path / "ids.bin")
    capi.host_index_file("ids", fa, ids_path)
    ids_bytes = open(ids_path, "rb").read()
    sub = random_sub(np.random.default_rng(1), ["a"], ids_bytes, 0, 1, 50, 256, 12, 15)
    good = str(tmp_path / "good.idx")
    IF.write(good, [sub])
    data = open(good, "rb").read()
    for name, blob in (("magic", b"\0" * 8 + data[8:]), ("cut", data[:len(data) // 2]), ("batch", data[:8] + (5).to_bytes(8, "little") + data[16:])):
        bad = str(tmp_path / f"{name}.idx")
        open(bad, "wb").write(blob)
        with pytest.raises(capi.WfmError):
            capi.host_index_file("rewrite", fa, str(tmp_path / "o.idx"), in_path=bad)
