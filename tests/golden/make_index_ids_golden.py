"""Generates tests/golden/index_ids_golden.bin: the id section of an index file as the REFERENCE's own
SequenceIdManager::exportIdMapping writes it (src/map/include/sequenceIds.hpp:101-115), through
oracle/_ref/libref_filter.so (built in place from /root/reference by oracle/Makefile), for the 137 names of
tests/test_index_file_cpu.py::write_names.  Run here (the reference is not on the GPU box):
    python tests/golden/make_index_ids_golden.py"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyfilter  # noqa: E402
from test_index_file_cpu import write_names  # noqa: E402

d = tempfile.mkdtemp()
fa = write_names(d)
out = os.path.join(HERE, "index_ids_golden.bin")
pyfilter.ref_export_ids(fa, out)
print(out, os.path.getsize(out), "bytes")
