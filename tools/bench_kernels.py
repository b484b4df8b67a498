#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on one synthetic batch (used for rocprofv3 --pmc passes)."""
import argparse, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd import ops, _lib
from matdeeplearn_amd.process import synthetic_bulk
import _ab; _ab.apply()      # (tools/_ab.py: MDL_HIP_LIB / MDL_OPS of the A/B scripts -> explicit calls)

# the library reads no environment any more: MDL_CG_EP = 0 / 2 here selects the backward edge pass through ops.K3_VARIANT
_ep = os.environ.get("MDL_CG_EP")
if _ep == "0":
    ops.K3_VARIANT = "per_wave"
elif _ep == "2":
    ops.K3_VARIANT = "edge_lane"
else:
    os.environ["MDL_CG_EP"] = "2"      # what the product runs at this size (labels of the counters below)

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=8192)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--which", default="fwd,bwd,rbf")
a = ap.parse_args()
dev = torch.device("cuda:0")
ds = synthetic_bulk(a.graphs, seed=0).to(dev)
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
b = ds.collate(np.arange(a.graphs), edge_dtype=dt)
C = a.dim
x = torch.randn(b.num_nodes, C, device=dev).to(dt).requires_grad_(True)
wf = (torch.randn(C, 2 * C + 50, device=dev) * 0.1).requires_grad_(True)
ws = (torch.randn(C, 2 * C + 50, device=dev) * 0.1).requires_grad_(True)
bf = torch.zeros(C, device=dev, requires_grad=True); bs = torch.zeros(C, device=dev, requires_grad=True)
print("N=%d E=%d" % (b.num_nodes, b.num_edges))
ev = {"fwd": [], "bwd": [], "bwd_node": []}
for it in range(a.iters + 2):
    ops.KERNEL_EVENTS = ev if it >= 2 else None
    out = ops.cgconv(x, None, b.edge_attr, wf, bf, ws, bs, "mean", csr=b.csr, split=os.environ.get("MDL_BK_SPLIT") == "1")
    if "bwd" in a.which:
        out.backward(torch.ones_like(out))
    if "rbf" in a.which:
        ops.rbf_expand(ds._dev["dist_norm"][: b.num_edges], out_dtype=dt)
ops.KERNEL_EVENTS = None
torch.cuda.synchronize()
for k, v in ev.items():
    if v:
        t = [s.elapsed_time(e) * 1e3 for s, e in v]
        print("%s: avg %.1f us  median %.1f us  min %.1f us" % (k, sum(t) / len(t), sorted(t)[len(t) // 2], min(t)))
try:
    import ctypes
    L = _lib.lib()
    if hasattr(L, "mdl_debug_read"):
        buf = (ctypes.c_longlong * 48)()
        L.mdl_debug_read(buf)
        v = list(buf)
        if hasattr(L, "mdl_debug_read_ep") and os.environ.get("MDL_CG_EP", "0") != "0":   # the edge-per-lane backward keeps
            buf2 = (ctypes.c_longlong * 48)()                                               # its counters in its own unit
            L.mdl_debug_read_ep(buf2)
            v[16:48] = list(buf2)[16:48]
        n = max(v[15], 1)
        if os.environ.get("MDL_CG_CB") == "1":
            cbn = ["data loads issue", "mfma chain", "gate+swaps", "reduce", "epilogue", "barrier", "commit (waits)", "idx issue"]
            print("cb fwd per-tile cycles (wave 0, %d tiles, total %d cyc/tile):" % (n, v[14] / n), {cbn[k]: round(v[k] / n) for k in range(8)})
        names = ["loop top (wait prefetch)", "commit+tsl", "issue x/prefetch", "pre0", "gate0", "reduce0", "pre1", "gate1", "reduce1", "-", "group prologue", "group epilogue"]
        print("fwd per-tile cycles (wave 0, %d tiles, total %d cyc/tile):" % (n, v[14] / n), {names[k]: round(v[k] / n) for k in range(12)}, "sum", round(sum(v[:12]) / n), "| shader clock %.2f GHz, wave-0 lifetime %.1f us/launch" % (v[14] / max(v[13], 1) * 0.1, v[13] / 100.0 / max(1, a.iters + 2)))
        n = max(v[31], 1)
        if os.environ.get("MDL_CG_CB_BWD") == "1":
            cbn = ["mfma chain", "dmv", "deriv+swaps", "reductions+dwe", "oob", "group flush", "commit+tables", "loads issue", "barrier"]
            print("cb bwd per-tile cycles (wave 0, %d tiles, total %d cyc/tile):" % (n, v[30] / n), {cbn[k]: round(v[16 + k] / n) for k in range(9)})
        names = ["loop top", "commit+tables", "issue loads", "pre", "dmv", "gate deriv", "pack", "reduce tgt", "reduce win", "dwe+rest", "prologue: zero+window base", "group epilogue", "prologue: loads+wait+gB"]
        if os.environ.get("MDL_CG_EP", "0") != "0":   # edge-per-lane kernel: cycles per ROUND of four tiles
            names = ["walk (take)", "A commit+tables", "A issue loads", "A mfma pair0", "A mfma pair1 || gate0", "A gathers + gate1", "-", "barrier 1", "B rest", "barrier 2", "B header+flush", "B operand wait+mfma", "B out-of-window"]
        print("bwd per-tile cycles (wave 0, %d tiles, total %d cyc/tile):" % (n, v[30] / n), {names[k]: round(v[16 + k] / n) for k in range(13)}, "sum", round(sum(v[16:29]) / n), "| shader clock %.2f GHz, wave-0 lifetime %.1f us/launch" % (v[30] / max(v[29], 1) * 0.1, v[29] / 100.0 / max(1, a.iters + 2)))
        if os.environ.get("MDL_CG_EP", "0") == "2":   # kernel 2: producer wave 0 above (first five slots), reducer wave 4 here; per ROUND
            n = max(v[47], 1)
            names = ["meta + group test", "group change (rest)", "operand reads + tf + R", "e reads + dwe", "window blocks", "round tail", "barrier",
                     "gc: before flush_r", "gc: flush_r", "gc: flush_w", "gc: roll", "round top: take", "round top: header + first request"]
            print("ep2 producer per-round cycles:", dict(zip(["wait top", "take+commit+slots", "issue ids/e", "mfma+gate+oow", "barrier"], [round(v[16 + k] / max(v[31], 1)) for k in range(5)])))
            print("ep2 reducer per-round cycles (%d rounds, total %d):" % (n, v[46] / n), {names[k]: round(v[32 + k] / n) for k in range(len(names))})
    if hasattr(L, "mdl_debug_life"):
        import numpy as np
        for which, nm in ((0, "fwd"), (1, "bwd")):
            buf = (ctypes.c_longlong * (4096 * 3))()
            (L.mdl_debug_life_ep if (which == 1 and hasattr(L, "mdl_debug_life_ep") and os.environ.get("MDL_CG_EP", "0") != "0") else L.mdl_debug_life)(buf, which)
            arr = np.array(list(buf), dtype=np.int64).reshape(4096, 3)
            arr = arr[arr[:, 1] > 0]
            if len(arr):
                t0 = arr[:, 0].min()
                st, en = (arr[:, 0] - t0) / 100.0, (arr[:, 1] - t0) / 100.0
                life = en - st
                print("%s waves %d: start us min/med/max %.1f %.1f %.1f | end us min/med/max %.1f %.1f %.1f | life us min/med/max %.1f %.1f %.1f | tiles min/med/max %d %d %d" % (
                    nm, len(arr), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), life.min(), np.median(life), life.max(),
                    arr[:, 2].min(), np.median(arr[:, 2]), arr[:, 2].max()))
                if os.environ.get("MDL_LIFE_DUMP"):
                    np.save(os.path.join(os.environ["MDL_LIFE_DUMP"], "life_%s.npy" % nm), np.stack([st, en, arr[:, 2]], 1))
                q = np.argsort(en)[-8:]
                print("   slowest waves:", [(int(k), round(float(st[k]), 1), round(float(en[k]), 1), int(arr[k, 2])) for k in q])
except Exception as e:
    print("no timing:", e)
