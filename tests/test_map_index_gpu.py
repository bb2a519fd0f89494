"""GPU parity test of the device-resident reference index: wfm_index_build
against the Python restatement of Sketch::build's index stage, on minmer intervals produced by
wfm_add_minmers for a small synthetic pangenome (repeats included so that the frequency filter
and the over-filtering safety check both fire)."""
import random

import numpy as np
import pytest

from oracle import map_index as MI
from wfmash_amd import synth

pytestmark = pytest.mark.gpu


def _pangenome(seed, n_seq, L, repeat=False):
    base = synth.random_dna(seed, L)
    if repeat:
        unit = synth.random_dna(seed + 1, 180)
        base = base[:L // 3] + unit * 60 + base[L // 3:]
    return [synth.mutate(base, 0.02, seed * 10 + i) for i in range(n_seq)]


def _check(gpu, seqs, k, w, s, max_freq):
    all_m = []
    for sid, sq in enumerate(seqs):
        m = gpu.add_minmers(sq, k, w, s, sid)
        all_m.append(m)
    mm = np.concatenate(all_m)
    ix = gpu.index_build(mm, max_freq)
    inf = ix.info()
    uh, po, pts, kept = ix.download()
    lookup, index, info = MI.build_index([(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in mm], max_freq)
    assert inf.n_windows == info["n_windows"] and inf.threshold == info["threshold"] and bool(inf.adjusted) == info["adjusted"]
    assert inf.n_kept == info["n_kept"] and inf.n_unique == info["n_unique"] and inf.filtered == info["filtered"]
    assert (np.diff(uh.astype(np.uint64)) > 0).all()  # unique hashes ascending -> binary-search lookup
    assert sorted(lookup.keys()) == [int(x) for x in uh]
    for u, hsh in enumerate(uh):
        got = [[int(p["pos"]), int(p["hash"]), int(p["seqId"]), int(p["side"])] for p in pts[po[u]:po[u + 1]]]
        assert got == lookup[int(hsh)], hex(int(hsh))
    assert [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in kept] == index
    # Sketch::build in one call: GPU hashing, thinned threaded winnowing, intervals straight to the device
    ix2, nw = gpu.index_build_sequences(seqs, k, w, s, threads=4, max_kmer_freq=max_freq)
    assert nw == len(mm)
    inf2 = ix2.info()
    assert [getattr(inf2, f) for f, _ in inf2._fields_] == [getattr(inf, f) for f, _ in inf._fields_]
    for a, b in zip(ix2.download(), (uh, po, pts, kept)):
        assert a.tobytes() == b.tobytes()
    ix2.free()
    ix.free()
    return inf


def test_index_build_plain(gpu):
    inf = _check(gpu, _pangenome(5, 6, 30000), 15, 256, 12, 0.0002)
    assert inf.n_unique > 1000 and inf.n_points >= 2 * inf.n_unique


def test_index_build_with_repeats_filters_and_adjusts(gpu):
    inf = _check(gpu, _pangenome(9, 5, 20000, repeat=True), 15, 256, 12, 0.0002)
    assert inf.filtered > 0
    inf2 = _check(gpu, _pangenome(9, 5, 20000, repeat=True), 15, 256, 12, 50.0)  # absolute count form of -F
    assert inf2.threshold == 50


def test_index_build_single_sequence_and_tiny(gpu):
    _check(gpu, [synth.random_dna(3, 5000)], 19, 64, 3, 0.0002)
    _check(gpu, [synth.random_dna(4, 300), synth.random_dna(5, 300)], 15, 100, 5, 0.5)
