"""The FASTA reader behind the map and align drivers (wfmash_amd/host/fasta.cpp), which stands in for
the reference's faigz/htslib layer (src/common/faigz.h:221-505): names in .fai order, lengths, and
faidx_reader_fetch_seq's inclusive-end substring fetches.  Random access (.fai; .gzi or a block scan for
BGZF) must return the same bytes as reading the whole file, for every container the reference accepts.
The expected values are plain Python slices of the sequences the files were written from."""
import gzip
import os
import random
import struct
import zlib

import pytest

from wfmash_amd import capi

pytestmark = pytest.mark.skipif(not os.path.exists(capi.LIB_PATH), reason="libwfmash_hip.so not built")


def make_seqs(seed=7):
    rng = random.Random(seed)
    lens = {"chrA#1#x": 300_017, "b": 1, "c desc ignored": 60, "d\tx=1": 61, "hapE#2#long": 131_072, "f": 59}
    return [(name, "".join(rng.choice("ACGTacgtN") for _ in range(n))) for name, n in lens.items()]


def fasta_text(seqs, width=60, eol="\n"):
    """FASTA bytes + the .fai lines samtools faidx would write for them."""
    out = bytearray()
    fai = []
    for hdr, s in seqs:
        out += f">{hdr}{eol}".encode()
        off = len(out)
        for i in range(0, len(s), width):
            out += (s[i:i + width] + eol).encode()
        fai.append(f"{hdr.split()[0]}\t{len(s)}\t{off}\t{width}\t{width + len(eol)}")
    return bytes(out), "\n".join(fai) + "\n"


def bgzf(data: bytes, block=0xff00, eof=True):
    """BGZF container (SAM spec 4.1) + the (compressed, uncompressed) start of every block."""
    out = bytearray()
    starts = []
    def put(chunk):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        out.extend(struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
                   struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    for i in range(0, len(data), block):
        starts.append((len(out), i))
        put(data[i:i + block])
    if eof:
        starts.append((len(out), len(data)))
        put(b"")
    return bytes(out), starts


def gzi_bytes(starts):
    body = starts[1:]  # the first block is implicit
    return struct.pack("<Q", len(body)) + b"".join(struct.pack("<QQ", c, u) for c, u in body)


def write(path, data, fai=None, gzi=None):
    with open(path, "wb") as f:
        f.write(data)
    if fai is not None:
        with open(str(path) + ".fai", "w") as f:
            f.write(fai)
    if gzi is not None:
        with open(str(path) + ".gzi", "wb") as f:
            f.write(gzi)


def check_store(path, seqs, expect_mode, rng, whole=True):
    table = capi.host_fasta(str(path)).splitlines()
    assert table[0] == expect_mode
    assert table[1:] == [f"{h.split()[0]}\t{len(s)}" for h, s in seqs]
    for hdr, s in seqs:
        name = hdr.split()[0]
        n = len(s)
        assert capi.host_fasta(str(path), name, 0, n - 1) == s
        assert capi.host_fasta(str(path), name, 0, n - 1, whole=whole) == s
        cases = [(0, 0), (n - 1, n - 1), (0, n + 100), (-5, 3), (n, n + 5), (59, 60), (60, 119), (58, 61)]
        cases += [tuple(sorted((rng.randrange(n), rng.randrange(n)))) for _ in range(12)]
        for a, b in cases:
            want = s[max(0, a):min(n, b + 1)] if a <= b else ""
            assert capi.host_fasta(str(path), name, a, b) == want, (name, a, b)
            assert capi.host_fasta(str(path), name, a, b, whole=whole) == want, (name, a, b, "whole")
    with pytest.raises(capi.WfmError):
        capi.host_fasta(str(path), "absent", 0, 1)


def test_plain_fasta_with_and_without_fai(tmp_path):
    seqs = make_seqs()
    rng = random.Random(1)
    data, fai = fasta_text(seqs)
    write(tmp_path / "a.fa", data, fai)
    check_store(tmp_path / "a.fa", seqs, "indexed", rng)
    write(tmp_path / "b.fa", data)
    check_store(tmp_path / "b.fa", seqs, "in-memory", rng)


def test_line_widths_and_crlf(tmp_path):
    seqs = make_seqs(3)
    rng = random.Random(2)
    for width, eol in ((61, "\n"), (60, "\r\n"), (1 << 20, "\n")):  # the last: every sequence on one line
        data, fai = fasta_text(seqs, width, eol)
        p = tmp_path / f"w{width}{len(eol)}.fa"
        write(p, data, fai)
        check_store(p, seqs, "indexed", rng)
        q = tmp_path / f"w{width}{len(eol)}.nofai.fa"
        write(q, data)
        check_store(q, seqs, "in-memory", rng)


def test_bgzf_with_gzi_without_gzi_and_streamed(tmp_path):
    seqs = make_seqs(5)
    rng = random.Random(3)
    data, fai = fasta_text(seqs)
    comp, starts = bgzf(data)
    assert gzip.decompress(comp) == data  # the writer above makes a valid multi-member gzip
    write(tmp_path / "gzi.fa.gz", comp, fai, gzi_bytes(starts[:-1]))  # index without the EOF block
    check_store(tmp_path / "gzi.fa.gz", seqs, "indexed", rng)
    write(tmp_path / "gzi_eof.fa.gz", comp, fai, gzi_bytes(starts))   # ... and with it
    check_store(tmp_path / "gzi_eof.fa.gz", seqs, "indexed", rng)
    write(tmp_path / "scan.fa.gz", comp, fai)                         # no .gzi: block headers are scanned
    check_store(tmp_path / "scan.fa.gz", seqs, "indexed", rng)
    small, st2 = bgzf(data, block=1000, eof=False)                     # many small blocks, no EOF marker
    write(tmp_path / "small.fa.gz", small, fai, gzi_bytes(st2))
    check_store(tmp_path / "small.fa.gz", seqs, "indexed", rng)
    write(tmp_path / "nofai.fa.gz", comp)                             # no .fai: streamed through zlib
    check_store(tmp_path / "nofai.fa.gz", seqs, "in-memory", rng)
    write(tmp_path / "plain.fa.gz", gzip.compress(data), fai)         # gzip that is not BGZF: no random access
    check_store(tmp_path / "plain.fa.gz", seqs, "in-memory", rng)


@pytest.mark.parametrize("align,threads", [(1, 2), (7, 5), (64, 16), (4096, 3), (2 << 20, 8)])
def test_long_sequences_filled_by_several_readers(tmp_path, monkeypatch, align, threads):
    """A chromosome goes into a block of its own that several threads fill, each its range of the bases (fasta.cpp:
    load_block).  Here every sequence is made 'long' (WFM_FASTA_BLOCK_MIN=1) and the readers' ranges are cut at multiples
    of `align` bases instead of 2 MB, so the cuts fall inside lines, on line ends, and past the end of short sequences; the
    bytes must be those of the one-reader path (the default thresholds) and of the file."""
    seqs = make_seqs(11)
    rng = random.Random(align * 31 + threads)
    monkeypatch.setenv("WFM_FASTA_BLOCK_MIN", "1")
    monkeypatch.setenv("WFM_FASTA_BLOCK_ALIGN", str(align))
    for width, eol in ((60, "\n"), (61, "\r\n"), (1 << 20, "\n")):
        data, fai = fasta_text(seqs, width, eol)
        p = tmp_path / f"blk{width}{len(eol)}.fa"
        write(p, data, fai)
        check_store(p, seqs, "indexed", rng, whole=threads)
    data, fai = fasta_text(seqs)
    small, st2 = bgzf(data, block=1000, eof=False)
    write(tmp_path / "blk.fa.gz", small, fai, gzi_bytes(st2))
    check_store(tmp_path / "blk.fa.gz", seqs, "indexed", rng, whole=threads)
    monkeypatch.setenv("WFM_FASTA_HUGE", "1")  # the huge-page hint changes nothing but the page size
    for hdr, sq in seqs[:2]:
        assert capi.host_fasta(str(tmp_path / "blk.fa.gz"), hdr.split()[0], 0, len(sq) - 1, whole=threads) == sq


def test_kept_store_is_reused_and_a_rewritten_file_is_read_afresh(tmp_path, monkeypatch):
    """A map call leaves its stores open for the align call that follows (fasta.hpp: keep_until_next); open_shared hands a
    kept store out again only while the file and its .fai are the ones it was opened on."""
    monkeypatch.setenv("WFM_FASTA_BLOCK_MIN", "1000")  # the long sequences of the set go through the block path
    p = tmp_path / "k.fa"
    a, b = make_seqs(21), make_seqs(22)
    for seqs in (a, b, a):  # the third round: same bytes and size as the first, a new file all the same
        data, fai = fasta_text(seqs)
        write(p, data, fai)
        for hdr, s in seqs:
            assert capi.host_fasta_shared(str(p), hdr.split()[0]) == s
        for hdr, s in seqs[:2]:  # again: served by the kept store
            assert capi.host_fasta_shared(str(p), hdr.split()[0]) == s
    capi.release_sequences()
    os.remove(p)
    with pytest.raises(capi.WfmError):
        capi.host_fasta_shared(str(p), "b")
    monkeypatch.setenv("WFM_FASTA_KEEP", "0")
    data, fai = fasta_text(a)
    write(p, data, fai)
    assert capi.host_fasta_shared(str(p), "b") == a[1][1]


def test_stale_index_is_an_error(tmp_path):
    seqs = make_seqs()
    data, fai = fasta_text(seqs)
    write(tmp_path / "cut.fa", data[:len(data) // 2], fai)
    with pytest.raises(capi.WfmError, match="does not match"):
        capi.host_fasta(str(tmp_path / "cut.fa"))
    write(tmp_path / "bad.fa", data, "chrA#1#x\tnot-a-number\n")
    with pytest.raises(capi.WfmError, match="malformed"):
        capi.host_fasta(str(tmp_path / "bad.fa"))
    with pytest.raises(capi.WfmError):
        capi.host_fasta(str(tmp_path / "missing.fa"))
