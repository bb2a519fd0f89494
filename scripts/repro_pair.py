import gzip, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wfmash_amd import capi
g = json.load(gzip.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "leaf_in_gap_pair.json.gz"), "rt"))
h = capi.Handle(0)
r = h.align([(g["pattern"].encode(), g["text"].encode())])[0]
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("WFM_")}, "status": r.status, "score": r.score, "want": g["score"], "cells": int(r.cells)}))
