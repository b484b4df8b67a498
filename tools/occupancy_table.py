#!/usr/bin/env python3
"""occupancy_table.py [file.hip ...]: CPU only.  Compiles the product's translation units with
-Rpass-analysis=kernel-resource-usage and prints, per kernel, registers / scratch / static LDS / the compiler's occupancy figure —
the listing behind the occupancy audit of DESIGN.md section 4 ("Round 4", item 14).  Dynamic LDS is chosen by the launchers and is
not in this listing: see the `lds` expressions next to each hipLaunchKernelGGL."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matdeeplearn_amd import _build  # noqa: E402

files = sys.argv[1:] or _build.sources()
pat = re.compile(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)")
for f in files:
    base = os.path.basename(f)
    cmd = [_build._hipcc()] + _build.FLAGS + _build.FILE_FLAGS.get(base, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rows = None, []
    for m in pat.finditer(out):
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[{"VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
                 "SGPRs Spill": "sspill", "VGPRs Spill": "vspill", "LDS Size [bytes/block]": "lds"}[k]] = v
    seen = set()
    print("== %s" % base)
    for r in rows:
        if r["name"] in seen:
            continue
        seen.add(r["name"])
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "").replace("mdl::", "")
        print("  %-72s vgpr %3s agpr %3s static-lds %6s waves/SIMD %s%s" % (
            name[:72], r.get("vgpr", "?"), r.get("agpr", "?"), r.get("lds", "?"), r.get("occ", "?"),
            ("   scratch %s B/lane (%s VGPRs spilled)" % (r.get("scratch"), r.get("vspill"))) if r.get("scratch", "0") != "0" else ""))
