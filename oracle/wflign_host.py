"""TEST INFRASTRUCTURE ONLY -- Python restatement of the reference's host-side align
logic around the WFA2-lib calls, used to check wfmash_amd/host/*.cpp.

Each function cites the reference code it follows (paths relative to
waveygang/wfmash).  The wavefront alignments themselves come from the C oracle
(oracle/wfa2p.c via pyoracle).  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import math
import re

import numpy as np

from . import pyoracle as O

_OP_RE = re.compile(r"(\d+)(\D)")

MIN_PATCH_LENGTH = 128        # wflign.cpp:169
MAX_ERODE_LENGTH = 4096       # wflign.cpp:170
MIN_CONSECUTIVE_MATCHES = 11  # wflign.cpp:171


def parse(cigar):
    return [(int(c), o) for c, o in _OP_RE.findall(cigar)]


def to_str(ops):
    return "".join(f"{c}{o}" for c, o in ops)


def compress(ops_bytes):
    """wfa_edit_cigar_to_string (wflign_swizzle.cpp:359-382) / compress_cigar (wflign.cpp:183-208)."""
    out = []
    prev, n = None, 0
    for b in ops_bytes:
        ch = chr(b)
        if ch == "M":
            ch = "="
        if ch == prev:
            n += 1
        else:
            if prev is not None:
                out.append(f"{n}{prev}")
            prev, n = ch, 1
    if prev is not None:
        out.append(f"{n}{prev}")
    return "".join(out)


def merge_adjacent_ops(c1, c2):
    """wflign.cpp:211-238."""
    if not c1:
        return c2
    if not c2:
        return c1
    a, b = parse(c1), parse(c2)
    if a and b and a[-1][1] == b[0][1]:
        return to_str(a[:-1] + [(a[-1][0] + b[0][0], a[-1][1])] + b[1:])
    return c1 + c2


def erode_short_matches_in_cigar(cigar, max_match_length=3, is_head=True):
    """wflign.cpp:19-106."""
    if len(cigar) < 6:
        return cigar
    ops = [list(x) for x in parse(cigar)]
    if len(ops) < 3:
        return cigar
    start, end = 1, len(ops) - 1
    if is_head:
        end = min(end, 3)
    else:
        start = max(start, len(ops) - 3)
    modified = False
    for i in range(start, end):
        is_match = ops[i][1] in "M=X"
        pair = (ops[i - 1][1], ops[i + 1][1])
        if (is_match and ops[i][0] <= max_match_length and pair in (("I", "D"), ("D", "I"))
                and ops[i - 1][0] > ops[i][0] and ops[i + 1][0] > ops[i][0]):
            ops[i - 1][0] += ops[i][0]
            ops[i + 1][0] += ops[i][0]
            ops[i][0] = 0
            modified = True
    if not modified:
        return cigar
    merged = []
    for c, o in ops:
        if c > 0:
            if merged and merged[-1][1] == o:
                merged[-1][0] += c
            else:
                merged.append([c, o])
    return to_str(merged)


def _consume(op, count, q, t):
    if op in "MX=":
        return q + count, t + count
    if op == "I":
        return q + count, t
    if op == "D":
        return q, t + count
    return q, t


def head_erosion(main_cigar):
    """wflign.cpp:241-276 -> (query_eroded, target_eroded, erode_end_pos)."""
    q = t = 0
    end_pos = 0
    found = False
    for m in _OP_RE.finditer(main_cigar):
        count, op = int(m.group(1)), m.group(2)
        if op == "=" and count >= MIN_CONSECUTIVE_MATCHES:
            found = True
        if found and q >= MIN_PATCH_LENGTH and t >= MIN_PATCH_LENGTH:
            break
        if q >= MAX_ERODE_LENGTH or t >= MAX_ERODE_LENGTH:
            break
        q, t = _consume(op, count, q, t)
        end_pos = m.end()
    return q, t, end_pos


def tail_erosion(ops):
    """wflign.cpp:331-364 -> (query_eroded, target_eroded, erode_start_idx)."""
    q = t = 0
    start_idx = len(ops)
    found = False
    for i in range(len(ops) - 1, -1, -1):
        count, op = ops[i]
        if op == "=" and count >= MIN_CONSECUTIVE_MATCHES:
            found = True
        if found and q >= MIN_PATCH_LENGTH and t >= MIN_PATCH_LENGTH:
            break
        if q >= MAX_ERODE_LENGTH or t >= MAX_ERODE_LENGTH:
            break
        q, t = _consume(op, count, q, t)
        start_idx = i
    return q, t, start_idx


def _merge_cigar_ops(cigar):
    """wflign_swizzle.cpp:7-37."""
    out = []
    for c, o in parse(cigar):
        if out and out[-1][1] == o:
            out[-1][0] += c
        else:
            out.append([c, o])
    return to_str(out)


def _seq_match(q, t, qs, ts, n):
    if qs < 0 or ts < 0 or qs + n > len(q) or ts + n > len(t):
        return False
    return q[qs:qs + n] == t[ts:ts + n]


def try_swap_start_pattern(cigar, query, target):
    """wflign_swizzle.cpp:217-260 (query_start = target_start = 0)."""
    ops = parse(cigar)
    if len(ops) < 2:
        return cigar
    (n, op1), (dlen, op2) = ops[0], ops[1]
    if op1 == "=" and op2 == "D" and _seq_match(query, target, 0, dlen, n):
        return _merge_cigar_ops(f"{dlen}D{n}=" + to_str(ops[2:]))
    return cigar


def try_swap_end_pattern(cigar, query, target):
    """wflign_swizzle.cpp:262-299; alignment_end_coords counts only '=' and 'D' and the
    verification accepts only '='/'D' CIGARs (wflign_swizzle.cpp:61-105,192-215)."""
    ops = parse(cigar)
    if len(ops) < 2:
        return cigar
    (dlen, op1), (n, op2) = ops[-2], ops[-1]
    if not (op1 == "D" and op2 == "="):
        return cigar
    end_q = sum(c for c, o in ops if o == "=")
    end_t = sum(c for c, o in ops if o in "=D")
    if not _seq_match(query, target, end_q - n, end_t - n - dlen, n):
        return cigar
    swapped = _merge_cigar_ops(to_str(ops[:-2]) + f"{n}={dlen}D")
    qp = tp = 0
    for c, o in parse(swapped):
        if o == "=":
            if qp + c > len(query) or tp + c > len(target) or query[qp:qp + c] != target[tp:tp + c]:
                return cigar
            qp += c
            tp += c
        elif o == "D":
            if tp + c > len(target):
                return cigar
            tp += c
        else:
            return cigar
    return swapped


def float2phred(prob):
    """wflign_patch.cpp:2726-2734."""
    if prob == 1:
        return 255.0
    p = -10 * math.log10(prob) if prob > 0 else float("inf")
    if p < 0 or p > 255:
        return 255.0
    return p


def _g(x):
    """Default iostream formatting of a floating value (6 significant digits)."""
    return "%g" % x


def write_alignment_paf(cigar, qname, qtotal, qoff, qlen, q_is_rev, tname, ttotal, toff,
                        mm_id, chain_id, chain_length, chain_pos,
                        min_identity=0.0, min_aln_len=32, min_block_identity=np.float32(0.1)):
    """wflign_patch.cpp:2611-2724 (+ trim_indels :139-223, process_compressed_cigar :226-283).
    Returns the line as the reference's writer emits it (tab separated, trailing tab) or None."""
    ops = parse(cigar)
    b, e = 0, len(ops)
    new_ref_start, new_query_start = toff, qoff
    while b < e and ops[b][1] in "ID":
        if ops[b][1] == "I":
            new_query_start += ops[b][0]
        else:
            new_ref_start += ops[b][0]
        b += 1
    if b < e:
        while e > b and ops[e - 1][1] in "ID":
            e -= 1
    core = ops[b:e]
    if not core:
        return None
    matches = sum(c for c, o in core if o in "M=")
    mism = sum(c for c, o in core if o == "X")
    ins = sum(1 for c, o in core if o == "I")
    ins_bp = sum(c for c, o in core if o == "I")
    dele = sum(1 for c, o in core if o == "D")
    del_bp = sum(c for c, o in core if o == "D")
    ref_len = matches + mism + del_bp
    q_len = matches + mism + ins_bp
    gi = matches / (matches + mism + ins + dele)
    bi = matches / (matches + mism + ins_bp + del_bp)
    if not (gi >= float(np.float32(min_identity)) and q_len >= min_aln_len and bi >= float(np.float32(min_block_identity))):
        return None
    if q_is_rev:
        q_start = qoff + (qlen - (new_query_start - qoff) - q_len)
        q_end = qoff + (qlen - (new_query_start - qoff))
    else:
        q_start, q_end = new_query_start, new_query_start + q_len
    fields = [qname, str(qtotal), str(q_start), str(q_end), "-" if q_is_rev else "+", tname, str(ttotal),
              str(new_ref_start), str(new_ref_start + ref_len), str(matches), str(max(ref_len, q_len)),
              _g(float(round(float2phred(1.0 - bi)))),
              "gi:f:" + _g(gi), "bi:f:" + _g(bi), "md:f:" + _g(float(np.float32(mm_id)))]
    if chain_length > 0:
        fields.append(f"ch:Z:{chain_id}.{chain_length}.{chain_pos}")
    fields.append("cg:Z:" + to_str(core))
    return "\t".join(fields) + "\t"


def do_biwfa_alignment(query, target, target_avail, pen=None, disable_chain_patching=False):
    """wflign.cpp:108-431 up to (not including) the writer.  `query`/`target` are bytes
    (strand-adjusted, upper case); `target_avail` = target plus its tail padding, as the
    swizzle sees it through the NUL-terminated buffer.  Returns the final CIGAR or None."""
    rc, ops, score, _ = O.align_biwfa(target, query, pen)
    if rc != 0:
        return None
    main = compress(ops)
    if not disable_chain_patching:
        qe, te, end_pos = head_erosion(main)
        if qe > 3 or te > 3:
            rc, hops, _, _ = O.align_endsfree(target[:te], te, 0, query[:qe], qe, 0, pen)
            if rc == 0:
                head = erode_short_matches_in_cigar(compress(hops), 3, True)
                main = merge_adjacent_ops(head, main[end_pos:])
        cops = parse(main)
        qe, te, start_idx = tail_erosion(cops)
        if qe > 3 or te > 3:
            tq = query[len(query) - qe:]
            tt = target[len(target) - te:]
            rc, tops, _, _ = O.align_endsfree(tt, 0, te, tq, 0, qe, pen)
            if rc == 0:
                tail = erode_short_matches_in_cigar(compress(tops), 3, False)
                main = merge_adjacent_ops(to_str(cops[:start_idx]), tail)
    main = try_swap_start_pattern(main, query, target_avail)
    main = try_swap_end_pattern(main, query, target_avail)
    return main


# ---------------------------------------------------------------------------
# align::Aligner front half (computeAlignments.hpp:195-303, 582-723)
# ---------------------------------------------------------------------------
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def upper_valid_dna(s: bytes) -> bytes:
    """makeUpperCaseAndValidDNA (commonFunc.hpp:132-142)."""
    a = np.frombuffer(s.upper(), dtype=np.uint8).copy()
    ok = (a == 65) | (a == 67) | (a == 71) | (a == 84)
    a[~ok] = ord("N")
    return a.tobytes()


def revcomp(s: bytes) -> bytes:
    return s.translate(_COMP)[::-1]


def is_a_number(s):
    return bool(s) and all(c in "0123456789." for c in s) and s.count(".") < 2


def parse_mashmap_row(line, target_padding, query_padding):
    """computeAlignments.hpp:195-303."""
    tok = line.split()
    if len(tok) < 13:
        raise ValueError("invalid mapping record")
    idv = tok[12].split(":")
    mm_id = np.float32(float(idv[-1])) if is_a_number(idv[-1]) else np.float32(0.70)
    chain_id, chain_length, chain_pos = -1, 1, 1
    if len(tok) > 14:
        cv = tok[14].split(":")
        if len(cv) == 3 and cv[0] == "ch" and cv[1] == "Z":
            parts = cv[2].split(".")
            if len(parts) == 3:
                chain_id, chain_pos, chain_length = int(parts[0]), int(parts[1]), int(parts[2])
    row = dict(qId=tok[0], qStartPos=int(tok[2]), qEndPos=int(tok[3]), rev=(tok[4] != "+"), refId=tok[5],
               chain_id=chain_id, chain_length=chain_length, chain_pos=chain_pos, mm_id=mm_id)
    ref_len, query_len = int(tok[6]), int(tok[1])
    rs, re_ = int(tok[7]), int(tok[8])
    qs, qe = row["qStartPos"], row["qEndPos"]
    if target_padding > 0:
        rs = rs - target_padding if rs >= target_padding else 0
        re_ = re_ + target_padding if re_ + target_padding <= ref_len else ref_len
    if query_padding > 0:
        if chain_pos == 1:
            qs = qs - query_padding if qs >= query_padding else 0
        if chain_pos == chain_length:
            qe = qe + query_padding if qe + query_padding <= query_len else query_len
            row["qStartPos"], row["qEndPos"] = qs, qe
    if rs >= ref_len or re_ > ref_len:
        raise ValueError("coordinates exceed reference length")
    row["rStartPos"], row["rEndPos"] = rs, re_
    return row


def align_mapping_lines(lines, ref_seqs, query_seqs, target_padding=1000, query_padding=1000,
                        max_len_minor=128000, pen=None):
    """Whole align phase for in-memory FASTA dicts {name: bytes}.  Returns PAF lines (no newline)."""
    out = []
    for line in lines:
        if not line.strip():
            continue
        try:
            row = parse_mashmap_row(line, target_padding, query_padding)
        except ValueError:
            continue
        ref = ref_seqs[row["refId"]]
        qry = query_seqs[row["qId"]]
        ref_size, q_size = len(ref), len(qry)
        tail_pad = min(ref_size - row["rEndPos"], max_len_minor)
        target_avail = upper_valid_dna(ref[row["rStartPos"]:row["rEndPos"] + tail_pad])
        tlen = row["rEndPos"] - row["rStartPos"]
        target = target_avail[:tlen]
        q = upper_valid_dna(qry[row["qStartPos"]:row["qEndPos"]])
        if row["rev"]:
            q = revcomp(q)
        cigar = do_biwfa_alignment(q, target, target_avail, pen)
        if cigar is None:
            continue
        rec = write_alignment_paf(cigar, row["qId"], q_size, row["qStartPos"], len(q), row["rev"], row["refId"],
                                  ref_size, row["rStartPos"], row["mm_id"], row["chain_id"], row["chain_length"],
                                  row["chain_pos"])
        if rec is not None:
            out.append("\t".join(rec.split()))  # processMappingRecord re-tokenisation (computeAlignments.hpp:484-525)
    return out


_FASTA_CACHE = {}


def read_fasta(path):
    """{name: bytes} of a plain FASTA file (test infrastructure: the workers of a process pool read the sequences themselves)."""
    if path not in _FASTA_CACHE:
        seqs, name, parts = {}, None, []
        with open(path, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        seqs[name] = b"".join(parts)
                    name, parts = line[1:].split()[0].decode(), []
                else:
                    parts.append(line.strip())
        if name is not None:
            seqs[name] = b"".join(parts)
        _FASTA_CACHE[path] = seqs
    return _FASTA_CACHE[path]


def align_mapping_lines_fasta(args):
    """align_mapping_lines for one chunk of rows, sequences from a FASTA path: (path, lines) -> PAF lines.  Picklable by name, so that a
    spawned process pool can run it (a GPU test must not fork the process that holds the HIP runtime)."""
    path, lines = args
    seqs = read_fasta(path)
    return align_mapping_lines(lines, seqs, seqs)


# ---------------------------------------------------------------------------
# SAM writer (wflign_patch.cpp:2480-2609) and MD:Z (write_tag_and_md_string :2397-2478)
# ---------------------------------------------------------------------------
def md_string(cigar, target_start, target):
    """The reference scans ops keeping a (last_op, last_len) pair: every op except the final
    one goes through the 'previous op' branch, the final one through the closing branch."""
    ops = []
    for c, o in parse(cigar):
        if ops and ops[-1][1] == o:
            ops[-1][0] += c
        else:
            ops.append([c, o])
    out = ["MD:Z:"]
    t_off, l_md = target_start, 0
    for i, (n, op) in enumerate(ops):
        last = i == len(ops) - 1
        if not last:
            if op in "=M":
                l_md += n
                t_off += n
            elif op == "X":
                for j in range(n):
                    out.append(f"{l_md}{chr(target[t_off + j])}")
                    l_md = 0
                t_off += n
            elif op == "D":
                out.append(f"{l_md}^" + target[t_off:t_off + n].decode())
                l_md = 0
                t_off += n
        elif n:
            if op in "=M":
                out.append(str(n + l_md))
            elif op == "X":
                for j in range(n):
                    out.append(f"{l_md}{chr(target[t_off + j])}")
                    l_md = 0
                out.append("0")
            elif op == "I":
                out.append(str(l_md))
            elif op == "D":
                out.append(f"{l_md}^" + target[t_off:t_off + n].decode() + "0")
    return "".join(out)


def write_alignment_sam(cigar, qname, qoff, q_is_rev, tname, toff, mm_id, chain_id, chain_length, chain_pos,
                        query, target, emit_md_tag=False, no_seq=False,
                        min_identity=0.0, min_aln_len=32, min_block_identity=np.float32(0.1)):
    ops = parse(cigar)
    b, e = 0, len(ops)
    new_ref_start, new_query_start = toff, qoff
    while b < e and ops[b][1] in "ID":
        if ops[b][1] == "I":
            new_query_start += ops[b][0]
        else:
            new_ref_start += ops[b][0]
        b += 1
    if b < e:
        while e > b and ops[e - 1][1] in "ID":
            e -= 1
    core = ops[b:e]
    if not core:
        return None
    matches = sum(c for c, o in core if o in "M=")
    mism = sum(c for c, o in core if o == "X")
    ins = sum(1 for c, o in core if o == "I")
    ins_bp = sum(c for c, o in core if o == "I")
    dele = sum(1 for c, o in core if o == "D")
    del_bp = sum(c for c, o in core if o == "D")
    q_len = matches + mism + ins_bp
    gi = matches / (matches + mism + ins + dele)
    bi = matches / (matches + mism + ins_bp + del_bp)
    if not (gi >= float(np.float32(min_identity)) and q_len >= min_aln_len and bi >= float(np.float32(min_block_identity))):
        return None
    trimmed = to_str(core)
    p0 = new_query_start - qoff
    seq = "*" if no_seq else query[p0:p0 + q_len].decode()
    f = [qname, "16" if q_is_rev else "0", tname, str(new_ref_start + 1), _g(float(round(float2phred(1.0 - bi)))), trimmed,
         "*", "0", "0", seq, "*", f"NM:i:{mism + ins_bp + del_bp}", "gi:f:" + _g(gi), "bi:f:" + _g(bi),
         "md:f:" + _g(float(np.float32(mm_id)))]
    if chain_length > 0:
        f.append(f"ci:i:{chain_id}")
        f.append(f"ch:Z:{chain_id}.{chain_length}.{chain_pos}")
    if emit_md_tag:
        f.append(md_string(trimmed, 0, target))
    return "\t".join(f)
