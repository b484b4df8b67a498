#!/usr/bin/env python3
"""Where the cost model of mdl_cgconv_balance comes from (CPU only).

Input: the per-wave lifetimes of ONE launch of the edge-per-lane backward on the bench batch, as written by
    MDL_LIFE_DUMP=<dir> MDL_CG_EP=2 MDL_HIP_LIB=<timing build> python tools/bench_kernels.py --which bwd
(a -DMDL_CG_TIMING build; `life_bwd.npy`: [waves, 3] = start us, end us, rounds; four producer waves per workgroup).
The script rebuilds the same batch on the CPU, cuts it into the kernel's node ranges — `--balanced 0`: equal shares of
edges + nodes, `--balanced 1`: equal shares of the cost prefix with the given weights — and regresses the workgroups' end
times on the content of their ranges (edges, tiles, edges whose source lies >= T rows from the target, ...).

    python tools/fit_balance.py gpurun_out/life/life_bwd.npy --balanced 0
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd.process.dataset import synthetic_bulk  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("life")
ap.add_argument("--graphs", type=int, default=8192)
ap.add_argument("--balanced", type=int, default=0)
ap.add_argument("--far-w", type=int, default=5)
ap.add_argument("--far-t", type=int, default=48)
ap.add_argument("--zero-w", type=int, default=4)
ap.add_argument("--workgroups", type=int, default=256)
a = ap.parse_args()

ds = synthetic_bulk(a.graphs, seed=0)                      # the batch tools/bench_kernels.py builds
src, tgt = np.asarray(ds.src), np.asarray(ds.tgt)
ep, npx = np.asarray(ds.edge_ptr), np.asarray(ds.node_ptr)
gid = np.repeat(np.arange(len(npx) - 1), np.diff(ep))
gs, gt = src + npx[gid], tgt + npx[gid]
N = int(npx[-1])
rowptr = np.zeros(N + 1, np.int64)
np.add.at(rowptr, gt + 1, 1)
rowptr = np.cumsum(rowptr)
d = np.abs(gs - gt)
deg = np.diff(rowptr)
if a.balanced:
    farn = np.zeros(N, np.int64)
    np.add.at(farn, gt, (d >= a.far_t).astype(np.int64))
    key = np.r_[0, np.cumsum(4 * (deg + 1) + a.far_w * farn + np.where(deg == 0, a.zero_w, 0))]
else:
    key = rowptr + np.arange(N + 1)
W = a.workgroups
bounds = [0] + [int(np.searchsorted(key, key[N] * w // W, side="left")) for w in range(1, W)] + [N]
life = np.load(a.life)
end = life[:, 1].reshape(-1, 4).max(1)[:W]
print("end times us: min / p10 / median / p90 / p99 / max", np.percentile(end, [0, 10, 50, 90, 99, 100]).round(1))
feats, names = [], ["nodes", "edges", "tiles", "far16", "far32", "far48", "far96"]
for w in range(W):
    na, nb = bounds[w], bounds[w + 1]
    e0, e1 = rowptr[na], rowptr[nb]
    dd = d[e0:e1]
    cnt = np.bincount((gt[e0:e1] - na) // 32, minlength=max((nb - na + 31) // 32, 1))
    feats.append((nb - na, e1 - e0, np.maximum(1, (cnt + 31) // 32).sum(), (dd >= 16).sum(), (dd >= 32).sum(), (dd >= 48).sum(), (dd >= 96).sum()))
F = np.array(feats, float)
for k, nm in enumerate(names):
    c = np.corrcoef(F[:, k], end)[0, 1] if F[:, k].std() > 0 else float("nan")
    print("%-6s mean %9.1f  std %7.1f  corr with end time %+.3f" % (nm, F[:, k].mean(), F[:, k].std(), c))
for cols in ([1, 5], [1, 4, 5, 6], [2, 5, 6]):
    A = np.c_[F[:, cols], np.ones(W)]
    coef, *_ = np.linalg.lstsq(A, end, rcond=None)
    pred = A @ coef
    r2 = 1 - ((end - pred) ** 2).sum() / ((end - end.mean()) ** 2).sum()
    print([names[c] for c in cols], "coef (us per unit)", coef[:-1].round(4), "R2 %.3f  residual std %.2f us  max residual %.1f us" % (r2, (end - pred).std(), (end - pred).max()))
    if cols == [1, 5]:
        # (with equal shares the edge counts hardly vary between workgroups, so their coefficient is not determined by the fit:
        # the average time per unit of work is the yardstick for the far-edge surcharge)
        per_unit = end.mean() / (F[:, 0] + F[:, 1]).mean()
        print("   -> a far edge costs %.2f of an average edge's time on top: weight %.1f quarter units" % (coef[1] / per_unit, 4 * coef[1] / per_unit))
