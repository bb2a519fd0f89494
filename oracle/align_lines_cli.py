"""Test infrastructure: oracle/wflign_host.py over the rows of a mapping file, many rows side by side.
    python oracle/align_lines_cli.py FASTA MAPPING_PAF OUT [--procs N]
writes the oracle's aligned PAF lines (row order) to OUT.  A process of its own, so that a GPU test never forks the process that holds the
HIP runtime: the pool below is forked from this one, which holds nothing but the sequences."""
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wflign_host as W  # noqa: E402

_SEQS = None


def _chunk(lines):
    return W.align_mapping_lines(lines, _SEQS, _SEQS)


def main():
    global _SEQS
    fa, paf, out = sys.argv[1:4]
    procs = int(sys.argv[sys.argv.index("--procs") + 1]) if "--procs" in sys.argv else min(48, os.cpu_count() or 1)
    _SEQS = W.read_fasta(fa)
    lines = [l for l in open(paf).read().splitlines() if l]
    chunks = [lines[i:i + 8] for i in range(0, len(lines), 8)]
    if procs > 1 and len(chunks) > 1:
        with mp.get_context("fork").Pool(min(procs, len(chunks))) as pool:
            parts = pool.map(_chunk, chunks)
    else:
        parts = [_chunk(c) for c in chunks]
    with open(out, "w") as f:
        for p in parts:
            for l in p:
                f.write(l + "\n")


if __name__ == "__main__":
    main()
