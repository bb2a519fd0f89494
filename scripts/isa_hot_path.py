#!/usr/bin/env python
"""Histogram of the instructions on the common path of ONE score step of wfa_tile2_kernel<1024, false, true>, from the compiler's
assembly (hipcc -S --cuda-device-only): the text between two consecutive s_barrier of the unrolled body, following the fall-through path
and TAKING every forward `s_cbranch_vccz / s_cbranch_execz` (the branches that skip work no lane has: the special selects, the global-mirror
probe, stage 2, the wave tail).  Each vector instruction is priced with the issue cycles profiles/r6_valu_issue.md measured for its kind
(W = 4 column); the sum is the issue time of a wave's step and the weighted mean is what bench.py's `valu` peak uses.
usage: isa_hot_path.py tile2.s [step_index]"""
import re
import sys
from collections import Counter

asm = open(sys.argv[1]).read().split("\n")
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# the phase-1 multi-wave FAST kernel
start = next(i for i, l in enumerate(asm) if l.startswith("_ZN3wfm16wfa_tile2_kernelILi1024ELb0ELb1ELb0EEE") and l.rstrip().endswith(":") or (l.startswith("_ZN3wfm16wfa_tile2_kernelILi1024ELb0ELb1ELb0EEE") and ":" in l))
end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
body = asm[start:end]
bars = [i for i, l in enumerate(body) if l.strip() == "s_barrier"]
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
# 2-cycle kinds measured at 2.2 cycles per wave64 instruction and SIMD; everything else vector at 4.2 (profiles/r6_valu_issue.md)
FAST2 = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_mov_b32", "v_not_b32")
i = bars[which] + 1
stop = bars[which + 1]
hist = Counter()
cyc = 0.0
n_valu = n_salu = n_lds = n_other = 0
guard = 0
prev = ""
while i < stop and guard < 100000:
    guard += 1
    l = body[i].strip()
    i += 1
    if not l or l.startswith(";") or l.startswith(".") and not l.startswith(".LBB"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", l):
        continue
    op = l.split()[0]
    m = re.match(r"s_cbranch_(vccz|execz)\s+(\.LBB\d+_\d+)", l)
    if not m and prev.startswith("s_cmp_eq_u64") and prev.rstrip().endswith(", 0"):  # `if (mask)`: no lane has anything
        m = re.match(r"s_cbranch_(scc1)\s+(\.LBB\d+_\d+)", l)
    prev = l
    if m and labels.get(m.group(2), -1) > i:
        n_salu += 1; hist[op] += 1
        i = labels[m.group(2)]
        continue
    hist[op] += 1
    if op.startswith("v_"):
        n_valu += 1
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        is_dpp = "dpp" in op or "row_" in l or "wave_" in l
        cyc += 2.2 if (base in FAST2 and not is_dpp) else 4.2
    elif op.startswith("s_"):
        n_salu += 1
    elif op.startswith("ds_"):
        n_lds += 1
    else:
        n_other += 1
print(f"step {which}: lines {bars[which]}..{stop}: VALU {n_valu}  SALU {n_salu}  LDS {n_lds}  other {n_other}")
print(f"issue cycles of the vector instructions: {cyc:.0f} per wave and step = {cyc / max(1, n_valu):.2f} per instruction")
for k, v in hist.most_common():
    print(f"  {v:4d} {k}")
