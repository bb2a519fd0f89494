"""Generates tests/golden/filter_golden.json.gz: for every case of tests/filter_cases.py the mapping
PAF text the REFERENCE'S OWN post-processing code prints (oracle/_ref/libref_filter.so, built from
/root/reference by oracle/Makefile).  Inputs are regenerated from the seeds, so only expected
outputs are stored.  Run from the repository root: python tests/golden/make_filter_golden.py"""
import gzip
import json
import os
import sys
import tempfile

sys.path.insert(0, os.getcwd())
from oracle import pyfilter  # noqa: E402
from tests import filter_cases as FC  # noqa: E402
from wfmash_amd import capi  # noqa: E402

assert pyfilter.have_ref(), "build oracle/_ref first (make -C oracle)"
out = {}
with tempfile.TemporaryDirectory() as d:
    fa = FC.write_fai(d)
    for name, query, seed, over in FC.CASES:
        m = FC.make_mappings(name, query, seed, over)
        P = capi.map_default_params(**over)
        out[name] = {"subset": pyfilter.ref_filter("subset", m, fa, query, P)}
        if name in ("defaults", "n1", "n3", "n2_droprand", "overlap_half"):
            out[name]["onetoone"] = pyfilter.ref_filter("onetoone", m, fa, query, P)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "filter_golden.json.gz")
with gzip.GzipFile(path, "wb", mtime=0) as f:
    f.write(json.dumps(out, sort_keys=True).encode())
print("wrote", path, {k: len(v["subset"].splitlines()) for k, v in out.items()})
