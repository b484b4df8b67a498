#!/usr/bin/env python3
"""asm_blocks.py <file.s> <kernel-substring> [min_instrs]: basic blocks of one kernel in a hipcc -S listing with their
instruction mix (MFMA / transcendental / packed / other VALU / LDS / VMEM / SALU / waits) — the CPU-side view of where a
kernel's issue slots go (works without a GPU)."""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].strip() == "s_endpgm")


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op): return "trans"
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith("v_cvt"): return "v_cvt"
    if op.startswith("v_accvgpr"): return "v_acc"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "v_lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


blocks, cur, name = [], Counter(), "entry"
first = start
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blocks.append((name, first, cur))
        cur, name, first = Counter(), m.group(1), i
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    cur[cls(t.split()[0])] += 1
blocks.append((name, first, cur))
tot = Counter()
for n, f, c in blocks:
    tot.update(c)
    if sum(c.values()) >= min_n:
        print("%-14s line %6d  n=%5d  %s" % (n, f + 1, sum(c.values()), " ".join("%s=%d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))
print("TOTAL", sum(tot.values()), dict(tot))
