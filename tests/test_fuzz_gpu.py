"""A short round of the randomised parity campaigns (tests/fuzz_align.py, tests/fuzz_map.py) inside the suite: the tools stay runnable, and every
run of the suite holds a few hundred fresh-shaped problems against the oracles."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_align_campaign_two_rounds():
    sys.path.insert(0, HERE)
    import fuzz_align
    out = fuzz_align.run(rounds=2, items=160, seed=7, max_len=5000, quiet=True)
    assert out["problems"] == 320 and out["differ"] == 0 and out["failed"] == 0, out
    out = fuzz_align.run(rounds=2, items=120, seed=8, max_len=2500, pens=True, quiet=True)
    assert out["differ"] == 0 and out["failed"] == 0, out


def test_map_campaign_two_rounds():
    from oracle import pyfilter, pymap
    if not (pymap.have_ref() and pyfilter.have_ref()):
        pytest.skip("oracle/_ref is built from /root/reference")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_map.py"), "--rounds", "2", "--seed", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '"rounds_differ": 0' in r.stdout
