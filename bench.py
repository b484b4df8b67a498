#!/usr/bin/env python3
"""bench.py — edges/sec training CGCNN on (synthetic) bulk_data on N MI355X, one JSON line.

Workload = BASELINE.json configs[1]: "bulk_data CGCNN dim=64, 4 conv layers, bf16 on 1 MI355X".
The real Materials-Project bulk_data is not redistributable and absent, so the dataset is the
synthetic bulk-like recipe of SURVEY.md 8d (matdeeplearn_amd.process.synthetic_bulk, all 46,744 graphs).
A "step" is one full training step of the hot path on one batch per GPU: device-side batch assembly (incl. the
K1 RBF expansion), forward, l1 loss, backward, gradient all-reduce (N>1), fused AdamW.  The dataset
is resident in HBM before the timed region.  Batches are cut from the DeviceLoader stream of the TRAIN split
(rank r takes perm(seed, epoch)[r::world_size], the DistributedSampler contract of training.py:291-294), `--batch`
graphs per GPU per step whatever N is: weak scaling; value = edges over all ranks / max-over-ranks time.

Beside the headline the line carries (rank 0, N=1): `sustained` (>= 2 s of steps), `fp32_mode` (the parity-mode
step), `ref_batch_100` (the reference's batch size, config.yml:136), `cpu_baseline` (the oracle on the host cores on
the SAME batches) with the val-MAE / prediction deltas of the fp32 and bf16 HIP paths at the trained weights.
`--model schnet|megnet` switch to the cfg3 / cfg4 workloads (SchNet_demo on MOF-like graphs, MEGNet_demo on
bulk-like graphs) with their own roofline kernels; the default line is cfg2.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# Batch sizes differ by a few percent from step to step, so every step asks the caching allocator for slightly different
# block sizes; with exact-size blocks it keeps calling hipMalloc / hipFree (which synchronise) once the first few dozen
# steps have fragmented its pool.  Size classes (1/8-of-a-power-of-two steps) make the blocks reusable.
for _k in ("PYTORCH_ALLOC_CONF", "PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF"):
    os.environ.setdefault(_k, "roundup_power2_divisions:8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (model class, dataset generator, default graphs, default batch, model kwargs (reference config.yml), description)
    "cgcnn": ("CGCNN", "synthetic_bulk", 46744, 8192, None, "cfg2 bulk_data CGCNN"),
    "schnet": ("SchNet", "synthetic_mof", 4096, 1024,
               dict(dim1=100, dim2=100, dim3=150, cutoff=8, pre_fc_count=1, gc_count=4, post_fc_count=3), "cfg3 MOF_data SchNet_demo"),
    "megnet": ("MEGNet", "synthetic_bulk", 16384, 4096,
               dict(dim1=100, dim2=100, dim3=100, pre_fc_count=1, gc_count=4, gc_fc_count=1, post_fc_count=3), "cfg4 bulk_data MEGNet_demo"),
    # the two remaining members of the cfg5 ensemble, on surface-like slabs (config.yml:141-161, 206-225)
    "mpnn": ("MPNN", "synthetic_surface", 8192, 1024,
             dict(dim1=100, dim2=100, dim3=100, pre_fc_count=1, gc_count=4, post_fc_count=3), "cfg5 surface_data MPNN_demo"),
    "gcn": ("GCN", "synthetic_surface", 8192, 2048,
            dict(dim1=100, dim2=150, pre_fc_count=1, gc_count=4, post_fc_count=3), "cfg5 surface_data GCN_demo"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="cgcnn", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="graphs per GPU per step (default per workload; reference default is 100)")
    ap.add_argument("--graphs", type=int, default=0, help="synthetic dataset size (default: the workload's recipe)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                    help="compute mode: bf16 (headline), fp32 (exact parity mode), bf16x3 (fp32 storage, the CGConv products as three "
                         "bf16 MFMAs on split operands)")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--gc", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline only: skip sustained / fp32_mode / ref_batch_100 (used for the rocprofv3 pass, so that "
                         "the per-kernel averages of the trace are those of the measured workload)")
    ap.add_argument("--sustain-s", type=float, default=2.0)
    ap.add_argument("--settle-s", type=float, default=1.0,
                    help="MINIMUM seconds of untimed steps in front of the warm-up (0 = none); the settle phase then goes on until three "
                         "consecutive 8-step groups agree within 3 %% or --settle-cap-s is reached")
    ap.add_argument("--settle-cap-s", type=float, default=3.0, help="upper bound of the settle phase in seconds")
    ap.add_argument("--fp32-leg", action="store_true", help="with --no-extras: still time the fp32 (parity-mode) step (other_models legs)")
    ap.add_argument("--targets", default="composition", choices=["composition", "noise"],
                    help="composition: the standardised mean atomic number of a graph (a target the models can fit, so that val-MAE "
                         "deltas discriminate); noise: the N(0,1) targets of the SURVEY 8d recipe")
    ap.add_argument("--cpu-steps", type=int, default=1, help="timed oracle steps on the GPU run's own batches")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="cpu_baseline on torch.set_num_threads(os.cpu_count()) as SURVEY 8d words it (default: the faster of 16 / 64 "
                         "threads on a probe batch — the intra-op pool degrades far beyond the useful parallelism of these sizes); "
                         "with --cpu-steps 50 this is SURVEY 8d's protocol (minutes of CPU time: not the default)")
    ap.add_argument("--cpu-graphs", type=int, default=0,
                    help="graphs of each timed batch the oracle steps on (0 = whole batch; MPNN defaults to 16: the oracle "
                         "materialises the reference's E x C x C edge tensor)")
    ap.add_argument("--no-other-models", action="store_true",
                    help="skip the short schnet / megnet / gcn / mpnn legs the default cgcnn line carries (other_models)")
    ap.add_argument("--strong-steps", type=int, default=10, help="N > 1: timed steps of the strong-scaling leg (global batch fixed)")
    ap.add_argument("--force-strong", action="store_true",
                    help="run the strong-scaling leg at world size 1 too, as if the global batch were shared by two ranks (B / 2 "
                         "graphs per step): executes the N > 1 code path on a one-GPU box (tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the gradient exchange at world size 1 too (tests)")
    ap.add_argument("--ops", action="append", default=[], metavar="NAME=0|1",
                    help="dispatch option of matdeeplearn_amd.ops (ops.OPTIONS) for A/B runs; repeatable")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="assemble every batch on the compute stream at the start of its step instead of one step ahead on a side stream")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="HIP events around the roofline kernels on every n-th timed step (0 = 4 for >= 8 steps, else every step)")
    ap.add_argument("--collector", default="off", choices=["off", "freeze", "on"],
                    help="cyclic collector inside the timed region: off (round 5), freeze (gc.freeze() in front of the warm-up steps: "
                         "collections stay on but only walk objects created since), on (untouched)")
    ap.add_argument("--run-in", type=int, default=0,
                    help="diagnosis: this many of the warm-up steps are enqueued behind the barrier that opens the timed region "
                         "(config.device_ms_per_step then shows the K steps with a non-empty launch queue at the start)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--dataset-cache", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "mdl_bench_data"),
                    help="directory for the flat on-disk copy of the synthetic dataset ('' = always regenerate)")
    return ap.parse_args()


def algorithmic_bytes(E, N, C, G, s):
    """compulsory traffic of one CGConv layer — K2 (forward) and K3 (the WHOLE backward: edge pass + node
    kernel + gradient assembly; x, grad_out read and grad_x written once per node, e and x_src read and the gradient of
    x_src accumulated once per edge)."""
    fwd = E * (G * s + C * s + 4) + N * (2 * C * s + 4)
    bwd = E * (G * s + 2 * C * s + 4) + N * (3 * C * s + 4)
    return fwd, bwd


def k3_part_bytes(E, N, C, G, s):
    """Compulsory traffic of the two launches K3 is made of, each through its OWN interface:
    edge pass: per edge the e row, the x_src row and both indices; per node x as target, grad_out, the row pointer, and the
    by-target / by-source sums [N, 2C] written once each;  node kernel: per node r_tgt, r_src, x, grad_out in and grad_x out."""
    edge = E * (G * s + C * s + 8) + N * (2 * C * s + 4 + 2 * 2 * C * s)
    node = N * (2 * 2 * C * s + 3 * C * s)
    return edge, node


def batch_stream(loader, B):
    """Fixed-size batches of graph ids cut from the loader's epoch stream (epoch e = perm(seed, e)[rank::world])."""
    buf = np.zeros(0, dtype=np.int64)
    epoch = 0
    while True:
        while len(buf) < B:
            loader.set_epoch(epoch)
            buf = np.concatenate([buf, loader._order()])
            epoch += 1
        yield buf[:B]
        buf = buf[B:]


class _HostMark:
    """stand-in for a device event on a CPU 'device' (the gloo world-size-2 test of the N > 1 plumbing)"""
    def __init__(self):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _mark(dev):
    return torch.cuda.Event(enable_timing=True) if dev.type == "cuda" else _HostMark()


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def make_barrier(dev, use_dist):
    import torch.distributed as dist

    def _barrier():
        if use_dist:
            dist.barrier()
        _sync(dev)
    return _barrier


def all_reduce_scalars(vals, op, dev, use_dist):
    """one fp64 all-reduce (MIN / MAX / SUM) of a short list of host scalars; returns python floats"""
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
    return [float(v) for v in t.tolist()]


def settle_phase(step, stream, dev, use_dist, settle_s, cap_s):
    """Untimed real steps until the step time has CONVERGED on every rank: >= settle_s seconds AND three consecutive 8-step groups
    within 3 % of each other (cap cap_s).  The steps run through the SAME path as the timed ones — each assembles the batch that
    follows on the side stream (round 6: the settle steps used to collate on the compute stream, so the side stream's allocator
    pool had seen five batches when the timed region started and the driver's round-5 run paid three hipMalloc inside it)."""
    n, groups = 0, []
    if settle_s <= 0:
        return n, groups
    t_s = time.perf_counter()
    nxt = next(stream)
    while True:
        t_g = time.perf_counter()
        for _ in range(8):
            cur, nxt = nxt, next(stream)
            step(cur, False, nxt)
        n += 8
        _sync(dev)
        now = time.perf_counter()
        groups.append((now - t_g) / 8 * 1e3)
        last = groups[-3:]
        stable = len(last) == 3 and max(last) <= 1.03 * min(last)
        el, st = all_reduce_scalars([now - t_s, 1.0 if stable else 0.0], "min", dev, use_dist)
        if (el >= settle_s and st > 0) or el >= max(cap_s, settle_s):
            break
    return n, groups


def timed_region(step, step_ids, warmup, steps, dev, use_dist, ev_stride, barrier, run_in=0):
    """W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + device synchronisation on both sides; the
    MAX over ranks of the host-clock time and the SUM over ranks of the edges.  Returns a dict.
    `run_in` (diagnosis only, default 0): that many of the W warm-up steps are enqueued BEHIND the barrier, and the dict also
    carries `elapsed_dev` = device-event time from the start mark (recorded behind them) to the end mark — the K steps as the
    device sees them when its launch queue is not empty at the start."""
    total = warmup + steps
    run_in = max(0, min(run_in, warmup))
    for i in range(warmup - run_in):
        step(step_ids[i], False, step_ids[i + 1])
    barrier()
    for i in range(warmup - run_in, warmup):
        step(step_ids[i], False, step_ids[i + 1])
    on_gpu = dev.type == "cuda"
    ms_t0 = torch.cuda.memory_stats(dev) if on_gpu else {}
    marks = [_mark(dev)]
    marks[0].record()
    t0 = time.perf_counter()
    edges = nodes = 0
    ev_edges = ev_nodes = ev_steps = 0
    host_ms = []
    # HIP events around the roofline kernels on every `ev_stride`-th timed step: the event pairs are measurement overhead inside
    # the timed region (35-55 us per instrumented step), so the default run pays it on a quarter of its steps — and never on
    # the first one, which starts from an idle device and an empty launch queue (round 6)
    ev_phase = 1 if (ev_stride > 1 and steps > 1) else 0
    for i in range(warmup, total):
        k = i - warmup
        ev_on = k % ev_stride == ev_phase
        th = time.perf_counter()
        e, n = step(step_ids[i], ev_on, step_ids[i + 1] if i + 1 < total else step_ids[0])
        host_ms.append(round((time.perf_counter() - th) * 1e3, 3))
        edges += e
        nodes += n
        if ev_on:
            ev_edges, ev_nodes, ev_steps = ev_edges + e, ev_nodes + n, ev_steps + 1
        if k % 4 == 3 or i == total - 1:
            marks.append(_mark(dev))                 # (one event record per four steps: device-side step times)
            marks[-1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    ms_t1 = torch.cuda.memory_stats(dev) if on_gpu else {}
    by4, done = [], 0
    for a, b in zip(marks[:-1], marks[1:]):
        k = min(4, steps - done)
        by4.append(round(a.elapsed_time(b) / max(k, 1), 3))
        done += k
    elapsed_max, = all_reduce_scalars([elapsed], "max", dev, use_dist)
    edges_all, = all_reduce_scalars([edges], "sum", dev, use_dist)
    return {"elapsed": elapsed, "elapsed_max": elapsed_max, "edges": edges, "nodes": nodes, "edges_all": edges_all, "by4": by4,
            "elapsed_dev": marks[0].elapsed_time(marks[-1]) * 1e-3, "run_in": run_in,
            "ev_edges": ev_edges, "ev_nodes": ev_nodes, "ev_steps": ev_steps, "host_enqueue_ms": host_ms,
            "device_mallocs": int(ms_t1.get("num_device_alloc", 0) - ms_t0.get("num_device_alloc", 0))}


def strong_leg(step, s_ids, warmup, strong_steps, Bs, world, dev, use_dist, barrier):
    """N > 1: the GLOBAL batch fixed at the one-GPU size (Bs = B / N graphs per GPU and step); max-over-ranks time, sum of edges."""
    for i in range(warmup):
        step(s_ids[i], False, s_ids[i + 1])
    barrier()
    t1 = time.perf_counter()
    e_s = 0
    for i in range(warmup, len(s_ids)):
        e_s += step(s_ids[i], False, s_ids[i + 1] if i + 1 < len(s_ids) else None)[0]
    barrier()
    ts, = all_reduce_scalars([time.perf_counter() - t1], "max", dev, use_dist)
    es, = all_reduce_scalars([e_s], "sum", dev, use_dist)
    return {"value": round(es / ts, 1), "unit": "edges/s", "scaling": "strong", "steps": strong_steps,
            "ms_per_step": round(ts / strong_steps * 1e3, 4), "global_batch_graphs": Bs * world, "batch_graphs_per_gpu": Bs}


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
    from matdeeplearn_amd import models, ops, process
    from matdeeplearn_amd.process import DeviceLoader, split_data
    from matdeeplearn_amd.training import FlatDataParallel, make_optimizer

    if args.ops:
        ops.configure(**{kv.split("=")[0]: kv.split("=")[1] not in ("0", "false", "False") for kv in args.ops})
    use_dist = world > 1 or args.force_dist           # --force-dist exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cls_name, gen_name, n_graphs, B, mkw, wl_desc = WORKLOADS[args.model]
    n_graphs = args.graphs or n_graphs
    B = args.batch or B
    # weak scaling draws B graphs per GPU and step: every rank's partition of the train split (0.8 n / world) must hold at
    # least one full batch, or a "batch" would be spliced from several epochs of the partition and contain duplicate graphs
    # (at 8 x 8192 the reference-sized 46,744-graph recipe is too small).  The synthetic recipe is simply drawn longer.
    need = int(np.ceil(B * world / 0.8 * 1.02)) + 8
    grown = (not args.graphs) and n_graphs < need
    if grown:
        n_graphs = need
    if mkw is None:
        mkw = dict(dim1=args.dim, dim2=args.dim, pre_fc_count=1, gc_count=args.gc, post_fc_count=3)
    mkw = dict(mkw, pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True", act="relu",
               dropout_rate=0.0)

    # ---- data: identical synthetic dataset on every rank, resident in HBM; the reference's 0.8 / 0.05 / 0.15 split ----
    # generated once per machine, then memory-mapped from the flat on-disk format (process.GraphDataset.save_flat): the
    # synthetic recipe (a python loop over 46,744 structures) leaves the critical path of every later run
    t0 = time.time()
    cache = os.path.join(args.dataset_cache, "%s_%d_seed0.mdlflat" % (gen_name, n_graphs)) if args.dataset_cache else None
    if cache and os.path.exists(cache):
        ds = process.GraphDataset.load_flat(cache)
        data_src = "flat file"
    else:
        ds = getattr(process, gen_name)(n_graphs, seed=0)
        data_src = "generated"
        if cache and rank == 0:
            try:
                os.makedirs(args.dataset_cache, exist_ok=True)
                ds.save_flat(cache + ".tmp%d" % os.getpid())
                os.replace(cache + ".tmp%d" % os.getpid(), cache)
            except OSError:
                pass
    gen_s = time.time() - t0
    if args.targets == "composition":
        # a target the models can fit (the standardised mean atomic number of the graph; the one tests/test_gpu_workloads.py
        # uses): with the recipe's N(0,1) noise targets every model scores MAE ~ E|y| = 0.8 and val_mae_delta* could not
        # tell two models apart.  Speed is unaffected (same tensors, same kernels).
        zbar = np.add.reduceat(np.asarray(ds.z, dtype=np.float64), np.asarray(ds.node_ptr[:-1])) / np.diff(np.asarray(ds.node_ptr))
        ds.y = ((zbar - zbar.mean()) / zbar.std()).astype(np.float32).reshape(-1, 1)
    ds.to(dev)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tr_idx, va_idx, _ = split_data(len(ds), 0.8, 0.05, 0.15, seed=args.seed)
    B = min(B, len(tr_idx) // world)
    assert B >= 1 and B * world <= len(tr_idx), "a per-rank partition of the train split must hold one full batch"
    loader = DeviceLoader(ds, tr_idx, B, shuffle=True, seed=args.seed, rank=rank, world_size=world)
    stream = batch_stream(loader, B)
    total_steps = args.warmup + args.steps
    step_ids = [next(stream) for _ in range(total_steps)]

    # ---- model ------------------------------------------------------------------------------------
    torch.manual_seed(args.seed)
    model = getattr(models, cls_name)(ds, compute_dtype=args.dtype, **mkw).to(dev)
    dp = FlatDataParallel(model, force=use_dist)
    opt = make_optimizer(model.parameters(), "AdamW", lr=0.002 * world)   # lr x world_size, training.py:388-389

    ktimes = {"cgcnn": {"fwd": [], "bwd": [], "bwd_node": [], "bwd_grads": []}, "schnet": {"gmr_fwd": [], "cfconv_fwd": [], "cfconv_bwd_h": [], "cfconv_bwd_w": []}, "megnet": {"edge_linear": []},
              "gcn": {"gmr_fwd": []}, "mpnn": {"nnconv_fwd": []}}[args.model]

    prefetch = not args.no_prefetch

    def make_step(model, dp, opt, dtype):
        pending = {}                                 # ids (bytes) -> handle of the batch being assembled one step ahead

        def step(ids, timed, next_ids=None):
            ahead = pending.pop(ids.tobytes(), None)
            if ahead is not None:
                batch = ds.take_ahead(ahead)
            else:
                batch = ds.collate(ids, edge_dtype=dtype, x_dtype=dtype)
            if next_ids is not None and prefetch:
                # batch k + 1 is assembled (K8 + K1) on a side stream while this step computes: every step still pays one
                # assembly inside the timed region, it just no longer sits alone on the device
                pending.clear()
                pending[next_ids.tobytes()] = ds.collate_ahead(next_ids, edge_dtype=dtype, x_dtype=dtype)
            dp.zero_grad()
            ops.KERNEL_EVENTS = ktimes if timed else None
            with ops.zero_arena(dev):               # one zero fill per step for the kernels' small accumulators
                out = model(batch)
                loss = ops.loss("l1_loss", out, batch.y)
                ops.backward(loss)
            ops.KERNEL_EVENTS = None
            if dp.reduce_grads_async(force=use_dist):
                dp.finish()
            opt.step()
            return batch.num_edges, batch.num_nodes
        return step

    step = make_step(model, dp, opt, cdt)

    barrier = make_barrier(dev, use_dist)

    model.train()
    # Settle phase (untimed, before the W warm-up steps): a fresh process sees a window of 100-200 ms, a few dozen steps
    # after its first launches, in which every kernel runs 2-4x slower, and a leg of 900 launches per step with varying batch
    # sizes (MEGNet) spends its first dozens of steps on first-use costs — hipBLASLt heuristics per GEMM shape, hipMalloc until
    # the caching allocator holds every size class.  Real steps run until the step time has converged (settle_phase).
    settle_steps, settle_groups = settle_phase(step, stream, dev, use_dist, args.settle_s, args.settle_cap_s)
    # Allocator high-water mark: the caching allocator reuses a block only for a request that fits it, so the first step whose
    # batch has more edges than every earlier one allocates that step's [E, F] activations anew (hipMalloc: 40-60 ms for the eight
    # 438-MB tensors of a SchNet step — seen as ONE group of four at 15-22 ms in a leg of 8.0).  The largest of the batches that
    # are about to be timed runs once, untimed, in front of the warm-up steps — assembled on the SIDE stream like every timed
    # batch (each stream has its own pool of blocks), then consumed on the compute stream.
    # No cyclic-garbage collection inside the timed region (like `timeit`): the host enqueues a step in about the time the device
    # needs for it, so a generation-2 pass over the process's objects — tens of ms — shows as ONE slow group of four steps (seen
    # in the SchNet leg at a fixed step index, in four of nine runs).  The collection runs HERE, in front of the warm-up steps: a
    # host pause right in front of the timed region would let the device drop to its idle clocks.
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    if args.collector == "off":
        gc.disable()
    elif args.collector == "freeze":
        # everything alive now moves to the permanent generation: later collections only walk the objects created since, so a
        # full pass is microseconds — and cyclic garbage that holds device tensors still gets freed (with the collector OFF
        # such tensors pile up and the caching allocator calls hipMalloc inside the timed region: `device_mallocs`)
        gc.freeze()
    if args.settle_s > 0 and hasattr(ds, "edge_ptr"):
        epg = np.diff(np.asarray(ds.edge_ptr))
        npg = np.diff(np.asarray(ds.node_ptr))
        big = {max(range(len(step_ids)), key=lambda k: int(epg[np.asarray(step_ids[k])].sum())),
               max(range(len(step_ids)), key=lambda k: int(npg[np.asarray(step_ids[k])].sum()))}
        for k in sorted(big):
            step(step_ids[0], False, step_ids[k])          # (assembles batch k on the side stream ...)
            step(step_ids[k], False, step_ids[0])          # (... and steps on it)
            settle_steps += 2
    # (prefetch: every step also starts the assembly of the batch that follows on a side stream — the first timed batch during
    # the last warm-up step, and the last timed step one more (unused) batch, so that the K timed steps contain K assemblies)
    ev_stride = args.event_stride or (4 if args.steps >= 8 else 1)
    tr = timed_region(step, step_ids, args.warmup, args.steps, dev, use_dist, ev_stride, barrier, run_in=args.run_in)
    if args.collector == "freeze":
        gc.unfreeze()
    if gc_was:
        gc.enable()
    elapsed, elapsed_max, edges_all = tr["elapsed"], tr["elapsed_max"], tr["edges_all"]
    edges, nodes, by4 = tr["edges"], tr["nodes"], tr["by4"]
    ev_edges, ev_nodes, ev_steps = tr["ev_edges"], tr["ev_nodes"], tr["ev_steps"]

    # ---- strong scaling beside it (N > 1): the GLOBAL batch fixed at the one-GPU size, B / N graphs per GPU ----
    strong = None
    if (world > 1 or args.force_strong) and args.strong_steps > 0:
        div = world if world > 1 else 2
        Bs = max(1, B // div)
        s_stream = batch_stream(DeviceLoader(ds, tr_idx, Bs, shuffle=True, seed=args.seed + 1, rank=rank, world_size=world), Bs)
        s_ids = [next(s_stream) for _ in range(args.warmup + args.strong_steps)]
        strong = strong_leg(step, s_ids, args.warmup, args.strong_steps, Bs, world, dev, use_dist, barrier)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream, timed region) ----
    s = 2 if args.dtype == "bf16" else 4
    G = ds.num_edge_features
    e_step, n_step = edges / args.steps, nodes / args.steps
    e_ev, n_ev = ev_edges / max(ev_steps, 1), ev_nodes / max(ev_steps, 1)       # per step, over the steps that carried events
    dur = {k: [a.elapsed_time(b) * 1e-3 for a, b in v] for k, v in ktimes.items()}   # seconds
    avg = {k: (sum(v) / len(v) if v else float("nan")) for k, v in dur.items()}
    tot = {k: sum(v) for k, v in dur.items()}
    if args.model == "cgcnn":
        ab_fwd, ab_bwd = algorithmic_bytes(e_ev, n_ev, mkw["dim1"], G, s)
        ab_edge, ab_node = k3_part_bytes(e_ev, n_ev, mkw["dim1"], G, s)
        # K3 as SURVEY 8d defines it = edge pass + node kernel + gradient assembly, timed TOGETHER (the per-launch sum of
        # the three event brackets) against the whole-backward bytes; the edge pass alone against its own bytes beside it
        parts = [k for k in ("bwd", "bwd_node", "bwd_grads") if dur.get(k)]
        if len(parts) > 1 and len({len(dur[k]) for k in parts}) == 1:
            dur["k3"] = [sum(t) for t in zip(*(dur[k] for k in parts))]
        else:
            dur["k3"] = list(dur.get("bwd", []))
        avg["k3"] = sum(dur["k3"]) / len(dur["k3"]) if dur["k3"] else float("nan")
        tot["k3"] = sum(dur["k3"])
        ab = {"fwd": ab_fwd, "k3": ab_bwd, "bwd": ab_edge, "bwd_node": ab_node}
        kname = {"fwd": "mdl_cgconv_fwd", "bwd": "mdl_cgconv_bwd_hb (edge pass of K3)", "bwd_node": "mdl_cgconv_bwd_node_h (node kernel of K3)",
                 "k3": "K3 = " + " + ".join({"bwd": "mdl_cgconv_bwd_hb", "bwd_node": "mdl_cgconv_bwd_node_h",
                                               "bwd_grads": "mdl_cgconv_assemble_grads"}[k] for k in parts)}
    elif args.model == "schnet":                     # K4a (aggregation only): E(2 F s + 8) + N(F s + 4)   (csrc/gather.hip)
        F_ = mkw["dim3"]
        ab = {"gmr_fwd": e_ev * (2 * F_ * s + 8) + n_ev * (F_ * s + 4),
              # K4 (fused forward, csrc/cfconv.hip), SURVEY 8d: E(G s + 4 + F s + 4) + N(2 F s + 4) — the training form also WRITES the two
              # activations the backward reads (2 E F s), which SURVEY's figure (filter recomputed in the backward) does not count
              "cfconv_fwd": e_ev * (G * s + 4 + F_ * s + 4) + n_ev * (2 * F_ * s + 4),
              # K4b (round 6: the backward with the filter recomputed): dh = the same kernel on the by-source CSR (the same bytes);
              # parameter gradients = one pass reading rbf + indices + cutoff and gathering the g and h rows: E(G s + 12 + 2 F s)
              "cfconv_bwd_h": e_ev * (G * s + 4 + F_ * s + 4) + n_ev * (2 * F_ * s + 4),
              "cfconv_bwd_w": e_ev * (G * s + 12 + 2 * F_ * s)}
        kname = {"gmr_fwd": "mdl_gather_mul_reduce", "cfconv_fwd": "mdl_cfconv_fwd (K4: filter network + cutoff + h[src] * W + segmented sum)",
                 "cfconv_bwd_h": "mdl_cfconv_fwd on the by-source CSR (K4b: dh, filter recomputed)",
                 "cfconv_bwd_w": "mdl_cfconv_bwd_w (K4b: dW1, db1, dW2, db2 of the filter network, filter recomputed; incl. its reduce launch)"}
    elif args.model == "gcn":                        # K4a with a scalar edge weight: E(F s + 8) + N(F s + 4)
        F_ = mkw["dim1"]
        ab = {"gmr_fwd": e_ev * (F_ * s + 8) + n_ev * (F_ * s + 4)}
        kname = {"gmr_fwd": "mdl_gather_mul_reduce"}
    elif args.model == "mpnn":                       # K7: N C d3 s (Y, read once per source node) + E (d3 + C) s   (csrc/nnconv.hip)
        C_, d3 = mkw["dim1"], mkw["dim3"]
        ab = {"nnconv_fwd": n_ev * C_ * d3 * s + e_ev * (d3 + C_) * s}
        kname = {"nnconv_fwd": "mdl_nnconv_msg_fwd"}
    else:                                            # K6: E(d s [e in] + d s [out] + 3 d s [gathered rows] + 12)
        d = mkw["dim3"]
        ab = {"edge_linear": e_ev * (5 * d * s + 12)}
        kname = {"edge_linear": "mdl_linear_gather_act"}
    have = [k for k in ab if dur.get(k)]
    dom = max(have, key=lambda k: tot[k]) if have else None
    if args.model == "cgcnn" and dur.get("k3"):
        dom = "k3"

    # HBM bytes per launch from PMC counters (separate rocprofv3 --pmc passes over tools/bench_kernels.py,
    # tools/gpu_pmc.sh -> profiles/hbm_traffic.json), scaled to this batch's edge count
    traffic, traffic_src = {}, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        import glob
        import hashlib
        hsh = hashlib.sha256()
        for f in sorted(glob.glob(os.path.join(ROOT, "matdeeplearn_amd/csrc/cgconv*")) + glob.glob(os.path.join(ROOT, "matdeeplearn_amd/csrc/rbf.hip"))):
            hsh.update(open(f, "rb").read())
        traffic_src = {"file": "profiles/hbm_traffic.json", "measured_at_E": tj.get("E"),
                       "kernel_sources_sha16": tj.get("kernel_sources_sha16"),
                       "stale": tj.get("kernel_sources_sha16") != hsh.hexdigest()[:16],
                       "note": "PMC pass over tools/bench_kernels.py (tools/gpu_pmc.sh), scaled to this batch's edge count; stale = the "
                               "conv kernel sources changed since that pass"}
        if args.model == "cgcnn" and mkw["dim1"] == 64 and args.dtype == "bf16":
            for k in ("fwd", "bwd", "bwd_node"):
                if "mdl_cgconv_" + k in tj:
                    traffic[k] = int(tj["mdl_cgconv_" + k]["bytes"] * e_ev / tj["E"])
            if "bwd" in traffic and "bwd_node" in traffic:
                traffic["k3"] = traffic["bwd"] + traffic["bwd_node"]
    except (OSError, ValueError, KeyError):
        pass

    def roof(k):
        ach = ab[k] / avg[k] / 1e9
        return {"kernel": kname[k], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(k),
                "traffic_source": traffic_src if traffic.get(k) is not None else None,
                "avg_launch_us": round(avg[k] * 1e6, 2), "launches": len(dur[k]),
                "events": "HIP event pairs on the launch stream, every %s timed step (%d of %d)"
                          % ("" if ev_stride == 1 else "%d-th" % ev_stride, ev_steps, args.steps),
                "algorithmic_bytes_per_launch": int(ab[k])}

    value = edges_all / elapsed_max
    metric = {"cgcnn": "edges/sec training CGCNN on bulk_data", "schnet": "edges/sec training SchNet on MOF_data",
              "megnet": "edges/sec training MEGNet on bulk_data", "mpnn": "edges/sec training MPNN on surface_data",
              "gcn": "edges/sec training GCN on surface_data"}[args.model]
    res = {
        "metric": metric,
        "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s %s, %s; synthetic graphs (SURVEY 8d recipe %s, %d graphs, r=8A, 12 NN + self loop), "
                               "train split 0.8 through the DeviceLoader partition, batch %d graphs/GPU"
                               % (wl_desc, " ".join("%s=%s" % kv for kv in sorted(mkw.items()) if kv[0].startswith(("dim", "gc", "post"))),
                                  args.dtype, gen_name, len(ds), B),
                   "batch_graphs_per_gpu": B, "edges_per_step_per_gpu": int(e_step), "nodes_per_step_per_gpu": int(n_step),
                   "parallelism": "dp%d" % world, "dataset_load_s": round(gen_s, 1), "dataset_source": data_src, "settle_steps": settle_steps,
                   "settle_ms_per_step_by_8": [round(g, 2) for g in settle_groups[-6:]], "targets": args.targets,
                   "ms_per_step_by_4": by4,
                   "device_mallocs": tr["device_mallocs"], "host_enqueue_ms_per_step": tr["host_enqueue_ms"],
                   "device_ms_per_step": round(tr["elapsed_dev"] / args.steps * 1e3, 4), "run_in": tr["run_in"],
                   "conv_kernel_share_of_step": round(sum(v for k, v in tot.items() if k != "k3") / max(ev_steps, 1) * args.steps / elapsed, 3)},
    }
    if strong is not None:
        res["strong_scaling"] = strong
    if grown:
        res["config"]["dataset_note"] = ("the recipe was drawn to %d graphs (reference-sized: %d) so that every rank's partition "
                                         "of the train split holds a full batch of distinct graphs" % (n_graphs, WORKLOADS[args.model][2]))
    # the WHOLE conv stack of a step against the HBM roofline (SURVEY 8d yardsticks): algorithmic bytes of every conv block,
    # forward + backward, divided by the step time — the per-kernel roofline above covers only the block's dominant kernel,
    # which for SchNet / MEGNet / MPNN is a sliver of the step (config.conv_kernel_share_of_step)
    L = mkw.get("gc_count", args.gc)
    if args.model == "cgcnn":
        stack = L * (ab_fwd + ab_bwd)
        conv = "K2 + K3 per layer, SURVEY 8d (252 + 390 B/edge/layer at C = 64, bf16, in-degree 13)"
    elif args.model == "schnet":                     # K4 CFConv forward: E(G s + 4 + F s + 4) + N(2 F s + 4); backward taken as 2x forward
        F_ = mkw["dim3"]
        stack = L * 3 * (e_step * (G * s + 4 + F_ * s + 4) + n_step * (2 * F_ * s + 4))
        conv = "SURVEY 8d K4 forward bytes x 3 per layer (forward + the two backward passes over the same operands)"
    elif args.model == "megnet":                     # K6 edge block E(4 d s + 8) + node block E d s + 3 N d s; backward taken as 2x forward
        d = mkw["dim3"]
        stack = L * 3 * (e_step * (4 * d * s + 8) + e_step * d * s + n_step * 3 * d * s)
        conv = "SURVEY 8d K6 edge + node block forward bytes x 3 per block"
    elif args.model == "mpnn":
        C_, d3 = mkw["dim1"], mkw["dim3"]
        stack = L * 3 * (n_step * C_ * d3 * s + e_step * (d3 + C_) * s)
        conv = "K7 forward bytes (N C d3 s + E (d3 + C) s) x 3 per layer"
    else:
        F_ = mkw["dim1"]
        stack = L * 3 * (e_step * (F_ * s + 8) + n_step * (F_ * s + 4))
        conv = "K4a forward bytes with a scalar edge weight x 3 per layer"
    step_s = elapsed_max / args.steps
    res["step_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_step": int(stack), "achieved": round(stack / step_s / 1e9, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(stack / step_s / 1e9 / HBM_PEAK_GBS, 4),
                            "convention": conv, "note": "conv-stack bytes over the WHOLE step time (assembly, dense layers, "
                            "BatchNorm, pooling, optimizer included in the time, not in the bytes)"}
    if dom is not None:
        res["roofline"] = roof(dom)
        if dom == "k3":
            res["roofline"]["parts_avg_launch_us"] = {k: round(avg[k] * 1e6, 2) for k in ("bwd", "bwd_node", "bwd_grads") if dur.get(k)}
            res["roofline"]["note"] = ("K3 of SURVEY 8d = the whole CGConv backward of a layer: the sum of the event brackets of its "
                                       "launches against SURVEY's whole-backward bytes (390 B/edge at C = 64, bf16)")
            res["roofline_edge_pass"] = roof("bwd")
            if dur.get("bwd_node"):
                res["roofline_node_kernel"] = roof("bwd_node")
            res["roofline_other"] = roof("fwd")
        else:
            other = [k for k in have if k != dom]
            if other:
                res["roofline_other"] = roof(other[0])
            if len(other) > 1:                       # SchNet: K4 forward, K4b dh pass, K4b parameter-gradient pass
                res["roofline_kernels"] = {k: {kk: vv for kk, vv in roof(k).items() if kk in ("kernel", "achieved", "frac", "avg_launch_us",
                                                                                             "launches", "algorithmic_bytes_per_launch")}
                                           for k in have}

    if world == 1 and not args.no_extras:
        from matdeeplearn_amd.training import GraphedStep

        def run_for(step_fn, it, min_s=None, n=None, ahead=False):
            """step_fn over batches from `it` for >= min_s seconds (or exactly n steps); (edges, steps, seconds, marks).
            No device synchronisation inside the loop: an idle gap lets the GPU drop its clocks, and the ramp back up
            costs tens of milliseconds (seen as 2 ms/step on every 16-step chunk that followed a sync).  The host is
            throttled by the steps themselves (pinned-ring events / the launch queue), so host wall-clock marks are a
            fair per-chunk reading and the final sync closes the measurement."""
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e_tot = k = 0
            marks = []
            nxt = next(it)
            while True:
                cur, nxt = nxt, next(it)
                e_tot += (step_fn(cur, nxt) if ahead else step_fn(cur))[0]
                k += 1
                if n is not None:
                    if k >= n:
                        break
                elif k % 16 == 0:
                    marks.append(time.perf_counter() - t1)
                    if marks[-1] >= min_s:
                        break
            torch.cuda.synchronize()
            return e_tot, k, time.perf_counter() - t1, marks

        # ---- sustained: >= sustain_s seconds of steps.  (a) the eager step of the timed region, (b) the same step as ONE
        # HIP-graph replay per batch (training.GraphedStep: static padded buffers, optimizer inside the graph) ----------
        ms0 = torch.cuda.memory_stats(dev)
        e_sus, n_sus, dt, marks = run_for(lambda ids, nxt: step(ids, False, nxt), stream, min_s=args.sustain_s, ahead=True)
        ms1 = torch.cuda.memory_stats(dev)
        eager = {"value": round(e_sus / dt, 1), "ms_per_step": round(dt / n_sus * 1e3, 4), "steps": n_sus,
                 "device_mallocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                 "ms_per_step_by_16": [round((b - a) / 16 * 1e3, 2) for a, b in zip([0.0] + marks[:-1], marks)]}
        res["sustained"] = {"eager": eager}
        try:
            opt_g = make_optimizer(model.parameters(), "AdamW", lr=0.002, capturable=True)
            gs = GraphedStep(ds, model, opt_g, B, compute_dtype=cdt, indices=tr_idx)
            for _ in range(3):
                gs.step(next(stream))
            e_sus, n_sus, dt, marks = run_for(gs.step, stream, min_s=args.sustain_s)
            res["sustained"].update({"value": round(e_sus / dt, 1), "unit": "edges/s", "steps": n_sus, "seconds": round(dt, 2),
                                     "ms_per_step": round(dt / n_sus * 1e3, 4), "mode": "hip-graph replay",
                                     "replays": gs.replays, "eager_fallback_steps": gs.eager_steps,
                                     "capacity": [gs.sb.n_cap, gs.sb.e_cap],
                                     "ms_per_step_by_16": [round((b - a) / 16 * 1e3, 2) for a, b in zip([0.0] + marks[:-1], marks)]})
        except Exception as exc:                                 # report, never hide: the eager figure stands in
            res["sustained"].update({"value": eager["value"], "unit": "edges/s", "ms_per_step": eager["ms_per_step"],
                                     "steps": eager["steps"], "mode": "eager", "graph_error": repr(exc)[:300]})

        # ---- the same training step at the reference's batch size (config.yml:136 batch_size 100) ----------------
        rb = 100
        rb_stream = batch_stream(DeviceLoader(ds, tr_idx, rb, shuffle=True, seed=args.seed), rb)
        run_for(lambda ids: step(ids, False), rb_stream, n=10)
        e_small, n_small, dt, _ = run_for(lambda ids: step(ids, False), rb_stream, n=100)
        res["ref_batch_100"] = {"eager": {"value": round(e_small / dt, 1), "ms_per_step": round(dt / n_small * 1e3, 4)},
                                "edges_per_step": int(e_small / n_small)}
        try:
            opt_s = make_optimizer(model.parameters(), "AdamW", lr=0.002, capturable=True)
            gs_s = GraphedStep(ds, model, opt_s, rb, compute_dtype=cdt, indices=tr_idx)
            run_for(gs_s.step, rb_stream, n=10)
            e_small, n_small, dt, _ = run_for(gs_s.step, rb_stream, n=400)
            res["ref_batch_100"].update({"value": round(e_small / dt, 1), "unit": "edges/s", "steps": n_small,
                                         "ms_per_step": round(dt / n_small * 1e3, 4), "mode": "hip-graph replay",
                                         "eager_fallback_steps": gs_s.eager_steps})
        except Exception as exc:
            res["ref_batch_100"].update({"value": res["ref_batch_100"]["eager"]["value"], "unit": "edges/s",
                                         "ms_per_step": res["ref_batch_100"]["eager"]["ms_per_step"], "mode": "eager",
                                         "graph_error": repr(exc)[:300]})

        # ... and through the reference's own job entry (training.train_regular = training.py:377-539: epochs, validation, scheduler):
        # the step replays by itself at this batch size (Training.graph_replay: auto); the eager job beside it
        if args.model == "cgcnn" and args.dtype == "bf16":
            try:
                from matdeeplearn_amd.training import train_regular
                sub = np.asarray(tr_idx)[:4096 + 512]
                tjob = dict(job_name="bench", seed=args.seed, save_model="False", write_output="False")
                tmp = dict(model="CGCNN", epochs=4, lr=0.002, batch_size=rb, optimizer="AdamW", optimizer_args={},
                           scheduler="ReduceLROnPlateau", scheduler_args={"mode": "min", "factor": 0.8, "patience": 10},
                           compute_dtype="bf16", **{k: v for k, v in mkw.items()})
                splits = (sub[:4096], sub[4096:4096 + 256], sub[4096 + 256:])
                thr = {}
                for mode in ("auto", "False"):
                    ttr = dict(target_index=int(getattr(ds, "target_index", 0)), loss="l1_loss", train_ratio=0.8, val_ratio=0.05,
                               test_ratio=0.15, verbosity=0, graph_replay=mode)
                    r = train_regular("cuda", 1, ds, tjob, ttr, tmp, splits=splits, edge_dtype=cdt, log=lambda *a: None)
                    h = r["history"][1:]
                    secs, st = sum(x["time"] for x in h), sum(-(-x["graphs"] // rb) for x in h)
                    thr[mode] = {"ms_per_step_incl_validation": round(secs / max(st, 1) * 1e3, 4), "steps": st,
                                 "value": round(sum(x["edges"] for x in h) / max(secs, 1e-9), 1), "unit": "edges/s",
                                 "mode": "hip-graph replay" if "replays" in h[-1] else "eager", "train_error": round(h[-1]["train"], 5)}
                res["ref_batch_100"]["through_train_regular"] = {"graph_replay_auto": thr["auto"], "graph_replay_off": thr["False"],
                                                                 "train_graphs": 4096, "val_graphs": 256, "epochs_timed": 3}
            except Exception as exc:
                res["ref_batch_100"]["through_train_regular"] = {"error": repr(exc)[:300]}

    # ---- the parity modes: same model, same batches, compute_dtype fp32 (exact products) and — CGCNN, whose conv kernels have
    # the form — bf16x3 (fp32 storage, the conv products as three bf16 MFMAs on (hi, lo)-split operands) ---------------------
    if world == 1 and args.dtype == "bf16" and (args.fp32_leg or not args.no_extras):
        # (bf16x3: split conv kernels for CGCNN at C = 64 / (96, 128]; for every model the Linears' weight gradients on split operands)
        for mode in ("fp32", "bf16x3"):
            torch.manual_seed(args.seed)
            m32 = getattr(models, cls_name)(ds, compute_dtype=mode, **mkw).to(dev)
            m32.train()
            dp32 = FlatDataParallel(m32)
            opt32 = make_optimizer(m32.parameters(), "AdamW", lr=0.002)
            step32 = make_step(m32, dp32, opt32, torch.float32)
            n32 = max(3, min(args.steps, 10))
            for i in range(2):
                step32(step_ids[i], False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e32 = 0
            for i in range(n32):
                e32 += step32(step_ids[args.warmup + i % args.steps], False)[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            res[mode + "_mode"] = {"value": round(e32 / dt, 1), "unit": "edges/s", "ms_per_step": round(dt / n32 * 1e3, 4), "steps": n32,
                                   "vs_bf16_step": round(dt / n32 / (elapsed_max / args.steps), 2)}
            del m32, dp32, opt32


    # ---- the other BASELINE configurations as short legs of the same run (cfg3 SchNet_demo, cfg4 MEGNet_demo, the cfg5 members):
    # one sub-process each, a dataset of two batches, a few timed steps, bf16 parity on a small held-out sample ----------
    if world == 1 and args.model == "cgcnn" and not args.no_extras and not args.no_other_models:
        res["other_models"] = other_models(args)

    # ---- CPU baseline: the oracle (pure-torch restatement of the reference path) on host cores ----
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args, ds, model, cls_name, mkw, step_ids[args.warmup:], va_idx)
    if use_dist:
        dist.destroy_process_group()
    print(json.dumps(res))


def other_models(args):
    """Short legs of the other workloads, each in its own process (fresh allocator, a failure cannot take the headline
    down): {model: {ms_per_step, value [edges/s], edges_per_step, roofline kernel + frac, val_mae_delta_bf16, ...}}."""
    import subprocess
    out = {}
    # cgcnn_dim100: the headline model at the reference's DEFAULT width (config.yml:123 dim1 = 100; the CGCNN member of cfg1 /
    # cfg5) on the headline batch — static 128-channel kernels on zero-padded rows
    for name in ("schnet", "megnet", "gcn", "mpnn", "cgcnn_dim100"):
        model, extra = (("cgcnn", ["--dim", "100"]) if name == "cgcnn_dim100" else (name, []))
        B = WORKLOADS[model][3]
        # 20 timed steps behind a settle phase that runs until the step time has converged (>= 0.5 s, three 8-step groups within
        # 3 %, cap 3 s): round 4's 6 steps behind a fixed 0.3 s timed a fresh box's first-use costs (driver: 54.4 ms/step for
        # the MEGNet leg against 19.3 on a warm box)
        cmd = [sys.executable, os.path.abspath(__file__), "--model", model, "--steps", "20", "--warmup", "3", "--settle-s", "0.5",
               "--settle-cap-s", "3.0", "--no-extras", "--fp32-leg", "--graphs", str(int(B * 1.25 / 0.8) + 64), "--cpu-steps", "0",
               "--dataset-cache", args.dataset_cache, "--seed", str(args.seed), "--no-other-models", "--targets", args.targets] + extra + [a for o in args.ops for a in ("--ops", o)]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
            if not line:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(line[-1])
            cb = j.get("cpu_baseline") or {}
            f32 = j.get("fp32_mode") or {}
            d16 = cb.get("val_mae_delta_bf16")
            out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                         "steps": j["steps"], "ms_per_step_by_4": j["config"].get("ms_per_step_by_4"),
                         "settle_steps": j["config"].get("settle_steps"),
                         "settle_ms_per_step_by_8": j["config"].get("settle_ms_per_step_by_8"),
                         "device_mallocs": j["config"].get("device_mallocs"),
                         "edges_per_step": j["config"]["edges_per_step_per_gpu"],
                         "batch_graphs": j["config"]["batch_graphs_per_gpu"], "dtype": j["dtype"],
                         "roofline_kernel": (j.get("roofline") or {}).get("kernel"), "roofline_frac": (j.get("roofline") or {}).get("frac"),
                         "conv_kernel_share_of_step": j["config"].get("conv_kernel_share_of_step"),
                         "step_roofline_frac": (j.get("step_roofline") or {}).get("frac"),
                         "fp32_mode": {"ms_per_step": f32.get("ms_per_step"), "value": f32.get("value")},
                         "bf16x3_mode": ({"ms_per_step": j["bf16x3_mode"].get("ms_per_step"), "vs_bf16_step": j["bf16x3_mode"].get("vs_bf16_step")}
                                         if j.get("bf16x3_mode") else None),
                         "val_mae_delta": cb.get("val_mae_delta"), "val_mae_delta_bf16": d16, "val_mae_delta_bf16x3": cb.get("val_mae_delta_bf16x3"),
                         "pred_max_rel_delta_bf16": cb.get("pred_max_rel_delta_bf16"), "val_graphs": cb.get("val_graphs"),
                         # north_star's bound is |dMAE| < 1e-5 against the CPU path: which mode meets it, at what speed
                         "tolerance": {"north_star_val_mae_delta": 1e-5,
                                       "met_by": ("bf16" if d16 is not None and d16 < 1e-5 else
                                                  "bf16x3" if cb.get("val_mae_delta_bf16x3") is not None and cb["val_mae_delta_bf16x3"] < 1e-5 else
                                                  "fp32" if cb.get("val_mae_delta") is not None and cb["val_mae_delta"] < 1e-5 else "none"),
                                       "bf16_stated": "see BASELINE.md section 5 (per-model bf16 tolerance)"},
                         "wall_s": round(time.time() - t0, 1)}
        except Exception as exc:                                  # report, never hide
            out[name] = {"error": repr(exc)[:300]}
    return out


def cpu_baseline(args, ds, gpu_model, cls_name, mkw, timed_ids, va_idx):
    """Times the oracle model (same architecture, fp32) on the host cores on the GPU run's OWN first timed batches
    (bounded: 1 warm-up + `--cpu-steps` steps), and checks parity of the HIP fp32 and bf16 paths at the trained weights
    on a held-out validation sample: val MAE and the largest prediction difference."""
    import copy
    from oracle import models as omodels
    from oracle import ops as oops
    from matdeeplearn_amd import models
    from matdeeplearn_amd.training import make_optimizer

    cores = os.cpu_count() or 1
    # the oracle NNConv materialises the reference's E x C x C edge tensor (40 KB per edge at C = 100): MPNN steps on 16 graphs
    cap = args.cpu_graphs or {"MPNN": 16}.get(cls_name, 0)
    if cap:
        timed_ids = [ids[:cap] for ids in timed_ids]
    val_cap = 1024 if not cap else 4 * cap
    if args.cpu_steps <= 0:
        val_cap = min(val_cap, 64 if cls_name != "MPNN" else 16)   # parity-only leg (other_models): a small held-out sample
    cds = copy.copy(ds)
    cds._dev = {}
    cds.to("cpu")
    rbf = lambda d: oops.rbf_expand(d, 0.0, 1.0, ds.num_edge_features, 0.2)
    state = {k: v.detach().cpu() for k, v in gpu_model.state_dict().items()}
    okw = {k: v for k, v in mkw.items()}

    def oracle():
        om = getattr(omodels, cls_name)(cds, **okw)
        om.load_state_dict(state)
        return om

    def one_step(om, opt, b):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = torch.nn.functional.l1_loss(om(b), b.y)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    # thread-pool size: torch's intra-op pool degrades when the thread count far exceeds the useful parallelism of these
    # tensor sizes; pick the faster of two sizes on a small probe batch, then time the identical batches with it
    out = {}
    probe = cds.collate(timed_ids[0][:512], rbf=rbf) if args.cpu_steps > 0 else None
    best_nt, best_t = min(cores, 16), None
    if args.cpu_all_cores:
        best_nt = cores
    for nt in (sorted({min(cores, t) for t in (16, 64)}) if (args.cpu_steps > 0 and not args.cpu_all_cores) else []):
        torch.set_num_threads(nt)
        om = oracle()
        om.train()
        opt = make_optimizer(om.parameters(), "AdamW", lr=0.002)
        one_step(om, opt, probe)
        t = one_step(om, opt, probe)
        if best_t is None or t < best_t:
            best_nt, best_t = nt, t
    torch.set_num_threads(best_nt)
    if args.cpu_steps > 0:
        om = oracle()
        om.train()
        opt = make_optimizer(om.parameters(), "AdamW", lr=0.002)
        nsteps = max(1, min(args.cpu_steps, len(timed_ids) - 1))
        batches = [cds.collate(timed_ids[i], rbf=rbf) for i in range(nsteps + 1)]
        one_step(om, opt, batches[0])                                  # warm-up on the GPU run's first timed batch
        edges, t_total = 0, 0.0
        for b in batches[1:]:
            t_total += one_step(om, opt, b)
            edges += b.num_edges
        out = {"value": round(edges / t_total, 1), "unit": "edges/s", "cores": best_nt, "host_cores": cores, "kind": "port",
               "sample": "%s: %d fp32 training step(s) of the oracle %s on %s of the GPU run's own timed batch(es) (%d graphs, %d edges) "
                         "after 1 warm-up step, %.1f s; threads: %s"
                         % ("1-step sample" if nsteps == 1 else "%d-step sample" % nsteps, nsteps, cls_name,
                            ("the first %d graphs" % cap) if cap else "all graphs", len(timed_ids[1]), edges, t_total,
                            "all %d host cores (--cpu-all-cores)" % cores if args.cpu_all_cores else
                            "%d of %d (the faster of 16 / 64 on a probe batch)" % (best_nt, cores))}

    # parity at the trained weights on held-out graphs: oracle (CPU fp32) vs HIP fp32 vs HIP bf16
    om = oracle()
    om.eval()
    ids = np.asarray(va_idx[:val_cap])
    with torch.no_grad():
        bc = cds.collate(ids, rbf=rbf)
        p_cpu = om(bc)
        mae_cpu = float(torch.nn.functional.l1_loss(p_cpu, bc.y))
        scale = float(p_cpu.abs().max()) + 1e-12
        modes = [("fp32", "fp32", torch.float32), ("bf16", "bf16", torch.bfloat16)]
        modes.insert(1, ("bf16x3", "bf16x3", torch.float32))     # (exact fp32 forward for the models without split conv kernels)
        for tag, cd, dt in modes:
            gm = getattr(models, cls_name)(ds, compute_dtype=cd, **mkw).to(ds.device)
            gm.load_state_dict(gpu_model.state_dict())
            gm.eval()
            bg = ds.collate(ids, edge_dtype=dt, x_dtype=dt)
            p = gm(bg)
            mae = float(torch.nn.functional.l1_loss(p, bg.y))
            out["val_mae_hip_" + tag] = mae
            out["val_mae_delta" + ("" if tag == "fp32" else "_" + tag)] = abs(mae - mae_cpu)
            out["pred_max_rel_delta_" + tag] = float((p.cpu() - p_cpu).abs().max()) / scale
    out["val_mae_oracle_cpu"] = mae_cpu
    out["val_graphs"] = int(len(ids))
    return out


if __name__ == "__main__":
    main()
