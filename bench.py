#!/usr/bin/env python3
"""bench.py — edges/sec training CGCNN on (synthetic) bulk_data on N MI355X, one JSON line.

Workload = BASELINE.json configs[1]: "bulk_data CGCNN dim=64, 4 conv layers, bf16 on 1 MI355X".
The real Materials-Project bulk_data is not redistributable and absent, so the dataset is the
synthetic bulk-like recipe of SURVEY.md 8d (matdeeplearn_amd.process.synthetic_bulk).  A "step" is
one full training step of the hot path on one batch per GPU: device-side batch assembly (incl. the
K1 RBF expansion), forward, l1 loss, backward, gradient all-reduce (N>1), fused AdamW.  The dataset
is resident in HBM before the timed region.  Weak scaling: every rank processes `--batch` graphs
per step; value = edges over all ranks / max-over-ranks time.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8192, help="graphs per GPU per step (reference default is 100)")
    ap.add_argument("--graphs", type=int, default=16384, help="synthetic dataset size (full recipe: 46744)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--gc", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ref-batch", action="store_true",
                    help="also time the step at the reference's batch size 100 (off by default so that the per-kernel "
                         "averages of a rocprofv3 trace of the default command are those of the measured workload)")
    ap.add_argument("--cpu-graphs", type=int, default=1024, help="graphs per CPU-baseline step")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def algorithmic_bytes(E, N, C, G, s):
    """compulsory traffic of one CGConv layer launch."""
    fwd = E * (G * s + C * s + 4) + N * (2 * C * s + 4)
    bwd = E * (G * s + 2 * C * s + 4) + N * (3 * C * s + 4)
    return fwd, bwd


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (matdeeplearn_amd has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from matdeeplearn_amd import models, ops
    from matdeeplearn_amd.process import synthetic_bulk
    from matdeeplearn_amd.training import FlatDataParallel, make_optimizer

    use_dist = world > 1 or os.environ.get("MDL_FORCE_DIST") == "1"   # the env var exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- data: identical synthetic dataset on every rank, resident in HBM -----------------------
    t0 = time.time()
    ds = synthetic_bulk(args.graphs, seed=args.seed)
    gen_s = time.time() - t0
    ds.to(dev)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    total_steps = args.warmup + args.steps
    rng = np.random.default_rng(1234 + rank)
    B = min(args.batch, len(ds))
    step_ids = [rng.choice(len(ds), size=B, replace=False) for _ in range(total_steps)]

    # ---- model ------------------------------------------------------------------------------------
    torch.manual_seed(args.seed)
    model = models.CGCNN(ds, dim1=args.dim, dim2=args.dim, pre_fc_count=1, gc_count=args.gc, post_fc_count=3,
                         pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                         act="relu", dropout_rate=0.0, compute_dtype=args.dtype).to(dev)
    dp = FlatDataParallel(model)
    opt = make_optimizer(model.parameters(), "AdamW", lr=0.002 * world)   # lr x world_size, training.py:388-389

    ktimes = {"fwd": [], "bwd": []}

    def step(ids, timed):
        batch = ds.collate(ids, edge_dtype=cdt, x_dtype=cdt)
        dp.zero_grad()
        ops.KERNEL_EVENTS = ktimes if timed else None
        with ops.zero_arena(dev):               # one zero fill per step for the kernels' small accumulators
            out = model(batch)
            loss = torch.nn.functional.l1_loss(out, batch.y)
            loss.backward()
        ops.KERNEL_EVENTS = None
        dp.reduce_grads(force=use_dist)
        opt.step()
        return batch.num_edges, batch.num_nodes

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    model.train()
    for i in range(args.warmup):
        step(step_ids[i], False)
    barrier()
    t0 = time.perf_counter()
    edges = nodes = 0
    for i in range(args.warmup, total_steps):
        e, n = step(step_ids[i], True)
        edges += e
        nodes += n
    barrier()
    elapsed = time.perf_counter() - t0

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    etot = torch.tensor([float(edges)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(etot, op=dist.ReduceOp.SUM)
    elapsed_max, edges_all = float(tmax), float(etot)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream, timed region) ----
    C, G, s = args.dim, ds.num_edge_features, (2 if args.dtype == "bf16" else 4)
    e_step, n_step = edges / args.steps, nodes / args.steps
    ab_fwd, ab_bwd = algorithmic_bytes(e_step, n_step, C, G, s)
    dur = {k: [a.elapsed_time(b) * 1e-3 for a, b in v] for k, v in ktimes.items()}   # seconds
    avg = {k: (sum(v) / len(v) if v else float("nan")) for k, v in dur.items()}
    tot = {k: sum(v) for k, v in dur.items()}
    dom = "bwd" if tot["bwd"] >= tot["fwd"] else "fwd"
    ab = {"fwd": ab_fwd, "bwd": ab_bwd}

    # HBM bytes per launch from PMC counters (separate rocprofv3 --pmc passes over tools/bench_kernels.py,
    # tools/gpu_pmc.sh -> profiles/hbm_traffic.json), scaled to this batch's edge count
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        if C == 64 and args.dtype == "bf16":
            for k in ("fwd", "bwd"):
                if "mdl_cgconv_" + k in tj:
                    traffic[k] = int(tj["mdl_cgconv_" + k]["bytes"] * e_step / tj["E"])
    except (OSError, ValueError, KeyError):
        pass

    def roof(k):
        ach = ab[k] / avg[k] / 1e9
        return {"kernel": "mdl_cgconv_%s" % k, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(k),
                "avg_launch_us": round(avg[k] * 1e6, 2), "launches": len(dur[k]),
                "algorithmic_bytes_per_launch": int(ab[k])}

    value = edges_all / elapsed_max
    res = {
        "metric": "edges/sec training CGCNN on bulk_data",
        "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "cfg2 bulk_data CGCNN dim1=dim2=%d, %d conv layers, post_fc 3, %s; synthetic bulk-like "
                               "graphs (SURVEY 8d recipe, %d of 46744 graphs, r=8A, 12 NN + self loop), batch %d "
                               "graphs/GPU" % (args.dim, args.gc, args.dtype, len(ds), B),
                   "batch_graphs_per_gpu": B, "edges_per_step_per_gpu": int(e_step), "nodes_per_step_per_gpu": int(n_step),
                   "parallelism": "dp%d" % world, "dataset_gen_s": round(gen_s, 1),
                   "conv_kernel_share_of_step": round((tot["fwd"] + tot["bwd"]) / elapsed, 3)},
        "roofline": roof(dom),
        "roofline_other": roof("fwd" if dom == "bwd" else "bwd"),
    }

    # ---- the same training step at the reference's batch size (config.yml:136 batch_size 100): launch bound
    if world == 1 and args.ref_batch:
        rb = 100
        ids_small = [rng.choice(len(ds), size=rb, replace=False) for _ in range(60)]
        for i in range(10):
            step(ids_small[i], False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e_small = 0
        for i in range(10, 60):
            e_small += step(ids_small[i], False)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        res["ref_batch_100"] = {"value": round(e_small / dt, 1), "unit": "edges/s", "ms_per_step": round(dt / 50 * 1e3, 4),
                                "edges_per_step": int(e_small / 50)}

    # ---- CPU baseline: the oracle (pure-torch restatement of the reference path) on host cores ----
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args, ds, model)
    if use_dist:
        dist.destroy_process_group()
    print(json.dumps(res))


def cpu_baseline(args, ds, gpu_model):
    """Times the oracle CGCNN (same architecture, fp32) on the host cores on a bounded sample of the
    same workload, and checks val-MAE parity of the HIP fp32 path at fixed weights."""
    import copy
    from oracle import models as omodels
    from oracle import ops as oops
    from matdeeplearn_amd import models
    from matdeeplearn_amd.training import make_optimizer

    cores = os.cpu_count() or 1
    cds = copy.copy(ds)
    cds._dev = {}
    cds.to("cpu")
    rbf = lambda d: oops.rbf_expand(d, 0.0, 1.0, ds.num_edge_features, 0.2)
    rng = np.random.default_rng(99)
    nb = min(args.cpu_graphs, len(cds))
    batches = [cds.collate(rng.choice(len(cds), size=nb, replace=False), rbf=rbf) for _ in range(args.cpu_steps + 1)]
    state = {k: v.detach().cpu() for k, v in gpu_model.state_dict().items()}
    best = None
    # torch's intra-op pool degrades badly when the thread count far exceeds the useful parallelism
    # of these tensor sizes: try a few pool sizes and report the fastest (threads used = "cores").
    for nt in sorted({min(cores, t) for t in (16, 64, cores)}):
        torch.set_num_threads(nt)
        torch.manual_seed(args.seed)
        om = omodels.CGCNN(cds, dim1=args.dim, dim2=args.dim, pre_fc_count=1, gc_count=args.gc, post_fc_count=3)
        om.load_state_dict(state)
        opt = make_optimizer(om.parameters(), "AdamW", lr=0.002)
        om.train()
        edges, t_total = 0, 0.0
        for i, b in enumerate(batches):
            t0 = time.perf_counter()
            opt.zero_grad()
            loss = torch.nn.functional.l1_loss(om(b), b.y)
            loss.backward()
            opt.step()
            dt = time.perf_counter() - t0
            if i > 0:                       # first step = warm-up
                edges += b.num_edges
                t_total += dt
            if t_total > 20.0:
                break
        if best is None or edges / t_total > best[0]:
            best = (edges / t_total, nt, edges, t_total)
    rate, nt, edges, t_total = best
    torch.set_num_threads(nt)
    out = {"value": round(rate, 1), "unit": "edges/s", "cores": nt, "host_cores": cores, "kind": "port",
           "sample": "%d fp32 training steps of the oracle CGCNN (dim %d, %d conv) on batches of %d synthetic graphs "
                     "(%d edges), %.1f s" % (args.cpu_steps, args.dim, args.gc, nb, edges, t_total)}
    # val-MAE parity at fixed weights: HIP fp32 path vs oracle on the same held-out sample
    om.load_state_dict(state)
    om.eval()
    gm = models.CGCNN(ds, dim1=args.dim, dim2=args.dim, pre_fc_count=1, gc_count=args.gc, post_fc_count=3,
                      compute_dtype="fp32").to(ds.device)
    gm.load_state_dict(gpu_model.state_dict())
    gm.eval()
    ids = rng.choice(len(cds), size=nb, replace=False)
    with torch.no_grad():
        bc = cds.collate(ids, rbf=rbf)
        mae_cpu = float(torch.nn.functional.l1_loss(om(bc), bc.y))
        bg = ds.collate(ids, edge_dtype=torch.float32)
        mae_gpu = float(torch.nn.functional.l1_loss(gm(bg), bg.y))
    out["val_mae_oracle_cpu"] = mae_cpu
    out["val_mae_hip_fp32"] = mae_gpu
    out["val_mae_delta"] = abs(mae_cpu - mae_gpu)
    return out


if __name__ == "__main__":
    main()
