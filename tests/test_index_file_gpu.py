"""`-W` / `-I` on the GPU path: the index file written by wfmh_map holds exactly what the device
index holds (parsed with the restatement of the reference's read code), mapping from the file gives the PAF of a
run that builds the index itself, also with several target subsets (`-b`)."""
import os

import numpy as np
import pytest

from oracle import map_index_file as IF
from wfmash_amd import capi
from test_map_paf_gpu import _pangenome, _write_fasta

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch", [0, 60000], ids=["one_subset", "b60k"])
def test_write_then_read_index(gpu, tmp_path, batch):
    seqs = _pangenome(57)
    fa = str(tmp_path / "pan.fa")
    _write_fasta(fa, seqs)
    idx = str(tmp_path / "pan.idx")
    base = dict(percentage_identity=0.9, auto_pct_identity=0, threads=4)
    if batch:
        base["index_by_size"] = batch
    direct = str(tmp_path / "direct.paf")
    s0 = capi.map_paf(gpu, fa, direct, params=capi.map_default_params(**base))
    assert s0.written > 0
    # -W: the index only
    sw = capi.map_paf(gpu, fa, str(tmp_path / "none.paf"), params=capi.map_default_params(index_file=idx, write_index=1, **base))
    assert sw.written == 0 and sw.index_windows == s0.index_windows and sw.subsets == s0.subsets
    subs = IF.parse(idx)
    assert len(subs) == s0.subsets and (s0.subsets > 1) == bool(batch)
    names = [n for n, s in seqs]
    assert [n for sub in subs for n in sub["names"]] == names
    for b, sub in enumerate(subs):
        assert (sub["batch_idx"], sub["total_batches"], sub["w"], sub["s"], sub["k"]) == (b, len(subs), 1000, s0.sketch_size, 15)
        assert sorted(i for _, i in sub["ids"]) == list(range(len(seqs)))
        # the same subset through the C ABI: the file holds the device index, record for record
        members = [(i, sq) for i, (n, sq) in enumerate(seqs) if n in sub["names"] and len(sq) >= 1000]
        ix, nw = gpu.index_build_sequences([sq for _, sq in members], 15, 1000, s0.sketch_size, seq_ids=[i for i, _ in members], threads=2)
        uh, po, pts, kept = ix.download()
        ix.free()
        assert sub["minmers"].tobytes() == kept.tobytes()
        assert sorted(sub["keys"]) == [int(x) for x in uh]
        where = {int(h): u for u, h in enumerate(uh)}
        for key, lst in zip(sub["keys"], sub["lists"]):
            u = where[key]
            assert lst.tobytes() == pts[po[u]:po[u + 1]].tobytes()
    # -I: mapping from the file
    from_file = str(tmp_path / "from_file.paf")
    over = dict(base)
    over.pop("index_by_size", None)  # the batch size comes from the file
    sr = capi.map_paf(gpu, fa, from_file, params=capi.map_default_params(index_file=idx, write_index=0, **over))
    assert open(from_file).read() == open(direct).read()
    assert sr.subsets == s0.subsets and 0 < sr.index_windows <= s0.index_windows  # the file holds the intervals that passed the frequency filter
    # a file made with other parameters is refused
    with pytest.raises(capi.WfmError, match="differ"):
        capi.map_paf(gpu, fa, str(tmp_path / "x.paf"), params=capi.map_default_params(index_file=idx, write_index=0, kmer_size=17, **over))
