"""TEST INFRASTRUCTURE ONLY -- the on-disk index of the map phase restated for the tests: a parser that follows
the reference's read side (Sketch::readIndex -> readSubIndexHeader, readParameters, readSketchBinary,
readPosListBinary, src/map/include/winSketch.hpp:677-737,834-935; SequenceIdManager::importIdMapping,
sequenceIds.hpp:117-212) and a writer that follows its write side (:569-660, sequenceIds.hpp:101-115).
PARITY UNPINNED for the sketch / position-list sections: winSketch.hpp needs htslib, which the image lacks, so
the reference cannot write a file here; the id section is pinned by the reference's own exportIdMapping
(oracle/_ref/libref_filter.so, tests/golden/index_ids_golden.bin)."""
import struct

import numpy as np

MAGIC = 0xDEADBEEFCAFEBABE
MINMER = np.dtype([("hash", "<u8"), ("wpos", "<i8"), ("wpos_end", "<i8"), ("seqId", "<i4"), ("strand", "<i2"), ("pad_", "<i2")])
POINT = np.dtype([("pos", "<i8"), ("hash", "<u8"), ("seqId", "<i4"), ("side", "i1"), ("pad_", "u1", (3,))])


def parse_ids(buf, off=0):
    (n,) = struct.unpack_from("<Q", buf, off); off += 8
    entries = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<Q", buf, off); off += 8
        name = bytes(buf[off:off + ln]).decode(); off += ln
        (sid,) = struct.unpack_from("<i", buf, off); off += 4
        entries.append((name, sid))
    (next_id,) = struct.unpack_from("<i", buf, off); off += 4
    return entries, next_id, off


def parse(path):
    """-> list of sub-indexes (dicts) in file order"""
    buf = memoryview(open(path, "rb").read())
    off, subs = 0, []
    while off < len(buf):
        magic, bidx, btotal, bsize, nnames = struct.unpack_from("<QQQqQ", buf, off); off += 40
        assert magic == MAGIC, hex(magic)
        names = []
        for _ in range(nnames):
            (ln,) = struct.unpack_from("<Q", buf, off); off += 8
            names.append(bytes(buf[off:off + ln]).decode()); off += ln
        ids_from = off
        entries, next_id, off = parse_ids(buf, off)
        ids_bytes = bytes(buf[ids_from:off])
        w, s, k = struct.unpack_from("<qii", buf, off); off += 16
        (n,) = struct.unpack_from("<Q", buf, off); off += 8
        minmers = np.frombuffer(buf, dtype=MINMER, count=n, offset=off).copy(); off += n * 32
        (nk,) = struct.unpack_from("<Q", buf, off); off += 8
        keys, lists = [], []
        for _ in range(nk):
            key, npts = struct.unpack_from("<QQ", buf, off); off += 16
            lists.append(np.frombuffer(buf, dtype=POINT, count=npts, offset=off).copy()); off += npts * 24
            keys.append(key)
        subs.append(dict(batch_idx=bidx, total_batches=btotal, batch_size=bsize, names=names, ids=entries, next_id=next_id,
                         ids_bytes=ids_bytes, w=w, s=s, k=k, minmers=minmers, keys=keys, lists=lists))
    return subs


def write(path, subs):
    """subs as parse() returns them (ids_bytes is written as it is; keys in the given order)"""
    with open(path, "wb") as f:
        for sub in subs:
            f.write(struct.pack("<QQQqQ", MAGIC, sub["batch_idx"], sub["total_batches"], sub["batch_size"], len(sub["names"])))
            for n in sub["names"]:
                f.write(struct.pack("<Q", len(n)) + n.encode())
            f.write(sub["ids_bytes"])
            f.write(struct.pack("<qii", sub["w"], sub["s"], sub["k"]))
            m = np.ascontiguousarray(sub["minmers"], dtype=MINMER)
            f.write(struct.pack("<Q", len(m)) + m.tobytes())
            f.write(struct.pack("<Q", len(sub["keys"])))
            for key, pts in zip(sub["keys"], sub["lists"]):
                p = np.ascontiguousarray(pts, dtype=POINT)
                f.write(struct.pack("<QQ", key, len(p)) + p.tobytes())
