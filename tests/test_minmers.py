"""addMinmers: the host winnowing stage against golden vectors generated from the
reference's own commonFunc.hpp (tests/golden/make_map_golden.py) -- CPU -- and the full
wfm_add_minmers (GPU hashing + host winnowing) on the GPU box.  Bit-exact, order included."""
import gzip
import json
import os
import random

import numpy as np
import pytest

from oracle import pymap
from wfmash_amd import capi, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "map_golden.json.gz")


def _as_list(m):
    return [[str(int(x["hash"])), int(x["wpos"]), int(x["wpos_end"]), int(x["strand"])] for x in m]


def _cpu_winnow(seq, k, w, s, seq_id):
    h, st = pymap.hash_kmers(seq, k)
    return capi.host_winnow(seq, k, w, s, seq_id, h, st)


def test_winnow_matches_reference_golden():
    gold = json.load(gzip.open(GOLD, "rt"))
    assert len(gold["minmers"]) >= 15
    for e in gold["minmers"]:
        seq = gold["seqs"][e["seq"]].encode()
        got = _cpu_winnow(seq, e["k"], e["w"], e["s"], 3)
        assert _as_list(got) == e["minmers"], (e["seq"], e["k"], e["w"], e["s"])
        assert all(int(x["seqId"]) == 3 for x in got)


@pytest.mark.skipif(not pymap.have_ref(), reason="reference build (oracle/_ref) only exists in the authoring container")
def test_winnow_matches_reference_live():
    rng = random.Random(2)
    for i in range(40):
        n = rng.choice([300, 1200, 4000, 9000])
        s = bytearray(synth.random_dna(500 + i, n))
        for _ in range(rng.randrange(0, 4)):
            p = rng.randrange(0, n)
            L = rng.randrange(1, 60)
            s[p:p + L] = (b"N" * L)[:max(0, min(L, n - p))]
        if rng.random() < 0.3:  # repeats: the lazy-heap quirks only show up on repetitive sequence
            unit = bytes(s[:rng.choice([7, 31, 150])])
            s = bytearray((unit * (n // len(unit) + 1))[:n])
            for _ in range(n // 50):
                s[rng.randrange(0, n)] = rng.choice(b"ACGT")
        if rng.random() < 0.2:
            s[:5] = b"NNACN"  # N inside the first k-1 bases (no initial scan in addMinmers)
        s = bytes(s[:n])
        k = rng.choice([15, 19])
        w = rng.choice([64, 256, 1000])
        sk = rng.choice([3, 12, 39])
        if n < w:
            continue
        ref = pymap.ref_add_minmers(s, k, w, sk, 5)
        got = _cpu_winnow(s, k, w, sk, 5)
        assert _as_list(got) == _as_list(ref), (i, n, k, w, sk)


@pytest.mark.gpu
def test_add_minmers_gpu_matches_golden(gpu):
    gold = json.load(gzip.open(GOLD, "rt"))
    for e in gold["minmers"]:
        seq = gold["seqs"][e["seq"]].encode()
        got = gpu.add_minmers(seq, e["k"], e["w"], e["s"], 3)
        assert _as_list(got) == e["minmers"], (e["seq"], e["k"], e["w"], e["s"])


@pytest.mark.gpu
def test_add_minmers_gpu_equals_cpu_stage_on_long_sequence(gpu):
    seq = synth.random_dna(31, 300000)
    got = gpu.add_minmers(seq, 15, 1000, 39, 0)
    exp = _cpu_winnow(seq, 15, 1000, 39, 0)
    assert len(got) == len(exp) > 5000
    assert (got == exp).all()


# ---- speculative chunked winnowing == one stream ----

def _chunk_cases():
    from wfmash_amd import synth
    base = synth.random_dna(901, 60000)
    unit7 = synth.random_dna(902, 7)
    unit400 = synth.random_dna(903, 400)
    yield "random", base
    yield "n_runs", base[:9000] + b"N" * 1500 + base[9000:20000] + b"n" + base[20000:31000] + b"N" * 40 + base[31000:]
    yield "n_at_boundaries", base[:4990] + b"NNNNNNNNNNNNNNNNNNNN" + base[5010:9999] + b"N" + base[10000:]
    yield "microsatellite", base[:8000] + unit7 * 3000 + base[8000:20000]                 # fewer than s distinct k-mers per window
    yield "tandem_repeat", base[:5000] + unit400 * 60 + base[5000:15000]                  # the same hashes stay in the sketch for 24 kb
    yield "palindromes", base[:3000] + (b"ACGT" * 2000) + base[3000:9000]
    yield "low_then_n", (b"A" * 5000) + b"N" * 100 + base[:12000] + (b"AC" * 4000)


def test_chunked_winnowing_equals_single_stream():
    import numpy as np
    from oracle import pymap
    from wfmash_amd import capi
    total_replays = 0
    for name, seq in _chunk_cases():
        for (k, w, s) in ((15, 1000, 39), (15, 500, 16), (19, 256, 5)):
            h, st = pymap.hash_kmers(capi_norm(seq), k)
            one = capi.host_winnow(seq, k, w, s, 3, h, st)
            for chunk in (4 * w + 17, 10000, 25000):
                got, replays = capi.host_winnow_chunked(seq, k, w, s, 3, h, st, chunk)
                assert replays >= 0, (name, k, w, s, chunk, "fell back to one stream: an expired pool entry took part in a refill")
                assert len(got) == len(one) and got.tobytes() == one.tobytes(), (name, k, w, s, chunk, replays)
                total_replays += replays
    # the speculation is expected to hold almost everywhere (replays are legal, just slow)
    assert total_replays <= 6, total_replays


def _thin_cases():
    yield from _chunk_cases()
    rng = random.Random(99)
    rnd = bytes(rng.choice(b"ACGT") for _ in range(30000))
    yield "n_in_first_kmers", b"ACGNTACGTTGCA" + rnd[:20000]
    yield "n_runs", rnd[:7000] + b"N" * 3000 + rnd[7000:9000] + b"N" * 40 + rnd[9000:20000]
    yield "short_tandem", rnd[:6000] + (rnd[100:137] * 300) + rnd[6000:14000]
    yield "long_period_repeat", rnd[:4000] + rnd[4000:5200] * 8 + rnd[5200:12000]
    yield "barely_a_window", rnd[:1003]


def test_thinned_winnowing_equals_single_stream():
    """the stream wfm_add_minmers_multi really feeds its workers: k-mers above the hash threshold are dropped
    unless a window may need them (selection restated on the host here; the GPU's is held against it in
    test_prefilter_gpu_matches_host_definition).  Winnowing the kept k-mers alone must give addMinmers' records."""
    total_kept = total = 0
    for name, seq in _thin_cases():
        for (k, w, s) in ((15, 1000, 39), (15, 500, 16), (19, 256, 5), (15, 1000, 78)):
            if len(seq) < w:
                continue
            h, st = pymap.hash_kmers(capi_norm(seq), k)
            one = capi.host_winnow(seq, k, w, s, 3, h, st)
            for c_factor in (3.0, 1.5, 0.5):
                for chunk in (0, 4 * w + 17, 25000):
                    got, kept, replays = capi.host_winnow_thinned(seq, k, w, s, 3, h, st, c_factor, chunk)
                    assert replays >= 0, (name, k, w, s, c_factor, chunk)
                    assert len(got) == len(one) and got.tobytes() == one.tobytes(), (name, k, w, s, c_factor, chunk, len(kept), len(h))
                if c_factor == 3.0 and (k, w, s) == (15, 1000, 39):
                    total_kept += len(kept)
                    total += len(h)
    # the point of the exercise: most k-mers never reach the host (these cases are repeat- and N-heavy on purpose)
    assert total_kept < 0.6 * total, (total_kept, total)


def test_device_winnower_model_equals_single_stream():
    """map_winnow.hip's control flow and capacities, run on the host over plain arrays (the kernel shares the code and
    swaps in wave-wide primitives): every chunking must give addMinmers' records, or hand the sequence back"""
    handed = total = 0
    for name, seq in _thin_cases():
        for (k, w, s) in ((15, 1000, 39), (15, 500, 16), (19, 256, 5), (15, 1000, 78)):
            if len(seq) < w:
                continue
            h, st = pymap.hash_kmers(capi_norm(seq), k)
            one = capi.host_winnow(seq, k, w, s, 3, h, st)
            for c_factor in (3.0, 1.5):
                for chunk in (0, 4 * w + 17, 9000, 25000, -(4 * w + 17), -7000):  # negative: with replays forced
                    got, why = capi.host_winnow_model(seq, k, w, s, 3, h, st, c_factor, chunk)
                    total += 1
                    if got is None:
                        handed += 1
                        assert name == "n_in_first_kmers" and why == 1 << 31, (name, k, w, s, c_factor, chunk, hex(why))
                        continue
                    assert len(got) == len(one) and got.tobytes() == one.tobytes(), (name, k, w, s, c_factor, chunk)
    assert handed * 8 < total, (handed, total)


@pytest.mark.skipif(not pymap.have_ref(), reason="reference build (oracle/_ref) only exists in the authoring container")
def test_device_winnower_model_matches_reference_live():
    rng = random.Random(12)
    done = 0
    for i in range(30):
        n = rng.choice([4000, 9000, 30000])
        s = bytearray(synth.random_dna(700 + i, n))
        for _ in range(rng.randrange(0, 4)):
            p = rng.randrange(0, n)
            L = rng.randrange(1, 80)
            s[p:p + L] = (b"N" * L)[:max(0, min(L, n - p))]
        if rng.random() < 0.4:
            unit = bytes(s[:rng.choice([5, 31, 150, 700])])
            a = rng.randrange(0, n // 2)
            rep = (unit * (n // len(unit) + 1))[:n // 3]
            s[a:a + len(rep)] = rep
            for _ in range(n // 200):
                s[rng.randrange(0, n)] = rng.choice(b"ACGT")
        s = bytes(s[:n])
        k = rng.choice([15, 19])
        w = rng.choice([256, 1000])
        sk = rng.choice([5, 23, 39])
        ref = pymap.ref_add_minmers(s, k, w, sk, 5)
        h, st = pymap.hash_kmers(capi_norm(s), k)
        got, why = capi.host_winnow_model(s, k, w, sk, 5, h, st, 3.0, rng.choice([4 * w + 3, 6000]))
        if got is None:
            continue
        done += 1
        assert _as_list(got) == _as_list(ref), (i, n, k, w, sk)
    assert done >= 20


def test_spread_sort_equals_std_sort_ties_included():
    """records tie under the reference's (wpos, wpos_end) order and std::sort's tie order is part of the output:
    the multi-threaded form must reproduce it exactly"""
    rng = np.random.default_rng(5)
    for n, span in ((300_000, 40_000), (1_500_000, 1_000_000), (700_000, 50)):
        recs = np.zeros(n, dtype=capi.MINMER_DTYPE)
        # nearly sorted with local disorder and many ties, like a sequence's records
        base = np.sort(rng.integers(0, span, n))
        recs["wpos"] = base + rng.integers(-3, 4, n)
        recs["wpos_end"] = recs["wpos"] + rng.integers(1, 4, n)
        recs["hash"] = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64)
        recs["seqId"] = 1
        one = capi.host_sort_records(recs, 1)
        assert (np.diff(one["wpos"]) >= 0).all()
        for threads in (2, 8, 16):
            assert capi.host_sort_records(recs, threads).tobytes() == one.tobytes(), (n, span, threads)


def capi_norm(seq: bytes) -> bytes:
    """upper-case / N-mask as the hashing kernel does (the oracle hashes what it is given)"""
    up = seq.upper()
    return bytes(c if c in b"ACGT" else ord("N") for c in up)


@pytest.mark.gpu
def test_add_minmers_multi_threaded_chunks_equal_single_calls(gpu, monkeypatch):
    """the production path: GPU hashing + worker pool + speculative chunks (chunk size shrunk so that
    every sequence is cut several times) against one wfm_add_minmers call per sequence"""
    seqs = [s for _, s in _chunk_cases()]
    single = [gpu.add_minmers(sq, 15, 256, 12, i) for i, sq in enumerate(seqs)]
    monkeypatch.setenv("WFM_WINNOW_CHUNK", str(64 * 256))
    multi = gpu.add_minmers_multi(seqs, 15, 256, 12, threads=8)
    assert len(multi) == len(single)
    for a, b in zip(multi, single):
        assert len(a) == len(b) and a.tobytes() == b.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("force", ["1", "2"])
def test_add_minmers_streamed_replay_paths(force):
    """failed speculations (1) and the one-stream fall-back (2) fetch their k-mers again from the device:
    forced for every chunk, in a fresh process (the switch is read once), the output must not change"""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from wfmash_amd import capi\n"
        "from test_minmers import _chunk_cases\n"
        "import os\n"
        "h = capi.Handle(0)\n"
        "seqs = [s for _, s in _chunk_cases()]\n"
        "single = [h.add_minmers(sq, 15, 256, 12, i) for i, sq in enumerate(seqs)]\n"
        "multi = h.add_minmers_multi(seqs, 15, 256, 12, threads=8)\n"
        "assert all(a.tobytes() == b.tobytes() for a, b in zip(multi, single))\n"
        "print('same', sum(len(a) for a in multi))\n")
    env = dict(os.environ, WFM_WINNOW_CHUNK=str(64 * 256), WFM_WINNOW_FORCE=force, WFM_DEBUG="1", WFM_WINNOW_DEVICE="0")  # the host's chunks
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "same" in r.stdout, r.stderr[-2000:]
    streamed = [l for l in r.stderr.splitlines() if "streamed through the pinned ring" in l]
    assert len(streamed) == 1, r.stderr[-2000:]
    if force == "1":
        assert "(0 replayed)" not in streamed[0], streamed[0]


@pytest.mark.gpu
def test_prefilter_gpu_matches_host_definition(gpu):
    """the device's selection of k-mers (scan / radix sort / window counts, map_prefilter.hip) against the
    definition restated on the host, and hashes/strands against the hash kernel's"""
    for name, seq in _thin_cases():
        for (k, w, s, c) in ((15, 1000, 39, 3.0), (19, 256, 5, 1.5), (15, 500, 16, 3.0)):
            if len(seq) < w:
                continue
            h, st = gpu.hash_kmers(seq, k)
            _, kept, _ = capi.host_winnow_thinned(seq, k, w, s, 3, h, st, c, 0)
            pos, ph, ps = gpu.prefilter_kmers(seq, k, w, s, c)
            assert pos.tobytes() == kept.tobytes(), (name, k, w, s, c, len(pos), len(kept))
            assert (ph == h[pos]).all() and (ps == st[pos]).all(), (name, k, w, s, c)


@pytest.mark.gpu
@pytest.mark.parametrize("dev_chunk,force", [("1041", "0"), ("16384", "0"), ("2100", "1")])
def test_add_minmers_multi_winnows_on_the_device(dev_chunk, force):
    """the production path of a thinned stream: one wave per speculative chunk (map_winnow.hip), boundary states compared and
    interval starts resolved on the device, closing sort on the device; against the REFERENCE'S OWN addMinmers
    (commonFunc.hpp:440-708, compiled in place: oracle/_ref/libref_map.so), one call per sequence.  The sequences the
    device may hand back are the ones the test names (an N among the first k-mers that the reference does not notice)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from wfmash_amd import capi\n"
        "from oracle import pymap\n"
        "from test_minmers import _thin_cases\n"
        "h = capi.Handle(0)\n"
        "seqs = [s for _, s in _thin_cases() if len(s) >= 1000]\n"
        "for (k, w, s) in ((15, 256, 12), (15, 1000, 39), (19, 500, 70)):\n"
        "    # the oracle: the reference's own addMinmers (commonFunc.hpp:440-708, oracle/_ref/libref_map.so)\n"
        "    assert pymap.have_ref(), 'oracle/_ref/libref_map.so did not travel'\n"
        "    single = [pymap.ref_add_minmers(sq, k, w, s, i) for i, sq in enumerate(seqs)]\n"
        "    multi = h.add_minmers_multi(seqs, k, w, s, threads=8)\n"
        "    bad = [i for i, (a, b) in enumerate(zip(multi, single)) if a.tobytes() != b.tobytes()]\n"
        "    assert not bad, (k, w, s, bad)\n"
        "    print('same', k, w, s, sum(len(a) for a in multi))\n")
    env = dict(os.environ, WFM_WINNOW_CHUNK=str(64 * 1000), WFM_WINNOW_DEV_CHUNK=dev_chunk, WFM_DEBUG="1", WFM_WINNOW_FORCE=force,  # force 1: replays
               WFM_WINNOW_DEV_MIN="0")  # (sequences this short go to the host's workers by default)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.count("same") == 3, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stderr.splitlines() if "winnowing on the device" in l]
    assert len(lines) == 3, r.stderr[-3000:]
    for l in lines:
        nseq = int(l.split("winnowing on the device:")[1].split("sequences")[0])
        back = int(l.split(";")[1].split("handed back")[0])
        assert nseq >= 9 and back <= 1, l
        if force == "1":
            assert int(l.split(" chunks replayed")[0].split()[-1]) > 10, l


@pytest.mark.gpu
def test_production_switch_device_winnower_against_the_reference():
    """m3 at DEFAULT thresholds (no WFM_WINNOW_* override): a 5.5 Mbp sequence is above WFM_WINNOW_DEV_MIN, so hashing,
    thinning, one wave per speculative chunk and the closing std::sort order all run on the device; a 0.6 Mbp sequence in
    the same call goes wherever the production switch sends it.  Held against the REFERENCE'S OWN addMinmers
    (commonFunc.hpp:440-708, oracle/_ref/libref_map.so), record for record, order included.  The long sequence carries
    what shapes the reference's quirks: N runs, lower case, a microsatellite, a long-period tandem repeat."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, '.')\n"
        "import numpy as np\n"
        "from wfmash_amd import capi, synth\n"
        "from oracle import pymap\n"
        "assert pymap.have_ref(), 'oracle/_ref/libref_map.so did not travel'\n"
        "b = bytearray(synth.random_dna(0x3a3, 5_500_000))\n"
        "b[1_000_000:1_020_000] = b'N' * 20_000\n"
        "b[2_000_000:2_030_000] = (bytes(b[100:107]) * 5000)[:30_000]\n"
        "b[3_000_000:3_048_000] = bytes(b[5000:6200]) * 40\n"
        "b[4_000_000:4_100_000] = bytes(b[4_000_000:4_100_000]).lower()\n"
        "b[4_500_000:4_500_001] = b'n'\n"
        "seqs = [bytes(b), synth.random_dna(0x3a4, 600_000)]\n"
        "h = capi.Handle(0)\n"
        "for (k, w, s) in ((15, 1000, 39), (15, 1000, 78)):\n"
        "    multi = h.add_minmers_multi(seqs, k, w, s, threads=8, cap=2_000_000)\n"
        "    for i, sq in enumerate(seqs):\n"
        "        ref = pymap.ref_add_minmers(sq, k, w, s, i, cap=2_000_000)\n"
        "        assert len(multi[i]) == len(ref) and multi[i].tobytes() == ref.tobytes(), (k, w, s, i, len(multi[i]), len(ref))\n"
        "    print('same', k, w, s, [len(a) for a in multi])\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("WFM_WINNOW") and k not in ("WFM_FINISH_DEVICE", "WFM_PREFILTER")}
    env["WFM_DEBUG"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.count("same") == 2, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stderr.splitlines() if "winnowing on the device" in l]
    assert len(lines) == 2, r.stderr[-3000:]
    for l in lines:
        nseq = int(l.split("winnowing on the device:")[1].split("sequences")[0])
        back = int(l.split(";")[1].split("handed back")[0])
        assert nseq >= 1 and back == 0 and "closing sort on the device" in l, l


def _raw_records(rng, n, span, w):
    """raw interval records as a winnower emits them: nearly ordered by their end, many ties, some empty, some longer than w"""
    recs = np.zeros(n, dtype=capi.MINMER_DTYPE)
    end = np.sort(rng.integers(1, span, n))
    length = rng.integers(0, 3 * w // 2, n)
    length[rng.random(n) < 0.05] = rng.integers(w, 6 * w, int((rng.random(n) < 0.05).sum()) or 1)[0]
    recs["wpos_end"] = end + rng.integers(0, 3, n)
    recs["wpos"] = np.maximum(recs["wpos_end"] - length, 0)
    recs["hash"] = rng.integers(0, 50 if span < 1000 else 2**62, n, dtype=np.int64).astype(np.uint64)
    recs["strand"] = rng.integers(-3, 4, n)
    recs["seqId"] = 7
    return recs


def test_sortlike_model_equals_std_sort():
    """map_finish.hip's restatement of std::sort -- the partitions of one recursion depth as lists, swaps and a cut,
    then a stable sort -- leaves every tie where the library leaves it"""
    rng = np.random.default_rng(11)
    for n, span in ((17, 5), (300, 40), (5000, 300), (200_000, 3000), (400_000, 5_000_000), (100_000, 3)):
        recs = np.zeros(n, dtype=capi.MINMER_DTYPE)
        base = np.sort(rng.integers(0, span, n))
        recs["wpos"] = np.maximum(base + rng.integers(-3, 4, n), 0)
        recs["wpos_end"] = recs["wpos"] + rng.integers(1, 4, n)
        recs["hash"] = rng.integers(0, 2**63, n, dtype=np.int64).astype(np.uint64)
        assert capi.host_sortlike_model(recs).tobytes() == capi.host_sort_records(recs, 1).tobytes(), (n, span)
    # a range that spends its depth budget (heap-sorted by the library in both)
    n = 200_000
    recs = np.zeros(n, dtype=capi.MINMER_DTYPE)
    i = np.arange(n)
    recs["wpos"] = np.where(i % 2 == 1, i, n - i)
    recs["wpos_end"] = recs["wpos"] + 1
    recs["hash"] = i.astype(np.uint64)
    assert capi.host_sortlike_model(recs).tobytes() == capi.host_sort_records(recs, 1).tobytes()


@pytest.mark.gpu
def test_finish_records_on_the_device_equal_the_host(gpu):
    """cut into pieces of w windows, strand signs, std::sort's order with its ties, de-duplication: the device's
    data-parallel form (sortlike_level_kernel + a stable radix sort) against the library itself on the host"""
    rng = np.random.default_rng(23)
    for n, span, w in ((0, 10, 100), (1, 10, 100), (16, 40, 10), (17, 40, 10), (3000, 200, 16), (250_000, 40_000, 1000), (1_200_000, 30_000_000, 1000),
                       (300_000, 900, 256)):
        raw = _raw_records(rng, n, span, w) if n else np.zeros(0, dtype=capi.MINMER_DTYPE)
        exp = capi.host_finish_records(raw, w)
        got, levels, heaps = gpu.finish_records(raw, w)
        assert len(got) == len(exp) and got.tobytes() == exp.tobytes(), (n, span, w, len(got), len(exp), levels, heaps)
    # the depth budget: ranges go to the library's heapsort
    n = 300_000
    raw = np.zeros(n, dtype=capi.MINMER_DTYPE)
    i = np.arange(n)
    raw["wpos"] = np.where(i % 2 == 1, i, n - i)
    raw["wpos_end"] = raw["wpos"] + 1 + (i % 3)
    raw["hash"] = (i * 2654435761 % 1000).astype(np.uint64)
    raw["strand"] = 1
    exp = capi.host_finish_records(raw, 1000)
    got, levels, heaps = gpu.finish_records(raw, 1000)
    assert got.tobytes() == exp.tobytes(), (levels, heaps)


def test_sortlike_model_on_arbitrary_keys():
    """property: for ANY multiset of keys in ANY order the restated partition steps leave what std::sort leaves (the
    derivation in map_finish.hip -- lists A / B, T, the cut -- does not lean on the shape of winnowing records)"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.tuples(st.integers(0, 40), st.integers(0, 3)), min_size=0, max_size=1500), st.integers(0, 2))
    def check(pairs, shape):
        n = len(pairs)
        recs = np.zeros(n, dtype=capi.MINMER_DTYPE)
        if n:
            a = np.array(pairs, dtype=np.int64)
            if shape == 1:
                a = a[np.lexsort((a[:, 1], a[:, 0]))]      # already sorted
            elif shape == 2:
                a = a[np.lexsort((a[:, 1], a[:, 0]))][::-1]  # descending
            recs["wpos"] = a[:, 0]
            recs["wpos_end"] = a[:, 0] + a[:, 1]
        recs["hash"] = np.arange(n, dtype=np.uint64)  # tells tied records apart
        assert capi.host_sortlike_model(recs).tobytes() == capi.host_sort_records(recs, 1).tobytes()

    check()


def test_device_winnower_model_on_arbitrary_sequences():
    """property: the kernel's control flow (host model) over ANY chunking of short, repeat- and N-rich sequences gives the
    one-stream records, or hands the sequence back for the one reason it may (an unnoticed N among the first k-mers)"""
    from hypothesis import given, settings, strategies as st

    unit = st.text(alphabet="ACGT", min_size=1, max_size=12)

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.one_of(st.text(alphabet="ACGT", min_size=20, max_size=400), st.tuples(unit, st.integers(2, 60)).map(lambda t: t[0] * t[1]),
                              st.integers(1, 30).map(lambda n: "N" * n)), min_size=3, max_size=25),
           st.sampled_from([(15, 64, 5), (11, 40, 3), (15, 128, 12)]), st.integers(1, 6), st.booleans())
    def check(parts, kws, chunk_windows, force):
        k, w, s = kws
        seq = "".join(parts).encode()
        if len(seq) < w + k:
            return
        h, st_ = pymap.hash_kmers(capi_norm(seq), k)
        one = capi.host_winnow(seq, k, w, s, 3, h, st_)
        chunk = chunk_windows * w + 7
        got, why = capi.host_winnow_model(seq, k, w, s, 3, h, st_, 1.5, -chunk if force else chunk)
        if got is None:
            assert why == 1 << 31, hex(why)
            return
        assert got.tobytes() == one.tobytes()

    check()
