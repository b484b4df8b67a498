#!/usr/bin/env python3
"""spill_audit.py [file.hip ...]: CPU only.  For every kernel of the product's translation units that spills: how many of its scratch
RELOADS sit inside a loop, and how many of those are followed — before any other instruction that could hide it — by a wait on the
vector memory counter.  A spill reload is a scratch load: it shares `vmcnt` with the kernel's global loads, so such a wait drains
every prefetch issued before it ("Round 5" item 8: a third of K4's time before it was found).  Kernels
without spills are not listed.  `profiles/r05_spill_audit.txt` is this tool's output for the round's last build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matdeeplearn_amd import _build  # noqa: E402

files = sys.argv[1:] or _build.sources()
with tempfile.TemporaryDirectory() as td:
    for f in files:
        base = os.path.basename(f)
        out = os.path.join(td, base + ".s")
        cmd = [_build._hipcc()] + _build.FLAGS + _build.FILE_FLAGS.get(base, []) + ["-S", "--cuda-device-only", "-o", out, f]
        if subprocess.run(cmd, capture_output=True, text=True).returncode != 0:
            print("== %s: does not compile to assembly on its own" % base)
            continue
        lines = open(out).read().split("\n")
        kernels, cur = [], None
        for n, l in enumerate(lines):
            m = re.match(r"^(_Z[0-9A-Za-z_]+):", l)
            if m:
                cur = {"name": m.group(1), "reloads": 0, "in_loop": 0, "drain": 0, "spills": 0}
                kernels.append(cur)
                in_loop = False
                continue
            if cur is None:
                continue
            if re.match(r"^\.LBB\d+_\d+:", l):
                in_loop = "in Loop" in l or "Loop Header" in l
            if "scratch_store" in l:
                cur["spills"] += 1
            if "scratch_load" in l:
                cur["reloads"] += 1
                if in_loop:
                    cur["in_loop"] += 1
                    for k in range(n + 1, min(n + 8, len(lines))):     # the wait in front of the reloaded register's first use
                        if "s_waitcnt" in lines[k] and "vmcnt" in lines[k]:
                            cur["drain"] += 1
                            break
                        if re.match(r"^\.LBB", lines[k]):
                            break
        rows = [k for k in kernels if k["reloads"] or k["spills"]]
        if not rows:
            continue
        print("== %s" % base)
        for k in rows:
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "").replace("mdl::", "")
            print("  %-70s spill stores %3d  reloads %3d  of them in a loop %3d  followed by a vmcnt wait %3d" % (
                name[:70], k["spills"], k["reloads"], k["in_loop"], k["drain"]))
