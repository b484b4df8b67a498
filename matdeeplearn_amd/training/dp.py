"""Data parallelism over graphs: one process per GPU, ONE all-reduce of ONE flat fp32 gradient
buffer per step (RCCL over xGMI; `gloo` on CPU for the test-suite).

Replaces torch DistributedDataParallel as the reference uses it
(/root/reference/matdeeplearn/training/training.py:227-237 ddp_setup, :263-266 DDP wrap with
find_unused_parameters=True).  Gradient payloads here are 0.45-17.5 MB, i.e. latency
bound on xGMI (7 links x ~153 GB/s): bucketing, the unused-parameter graph walk and the per-forward
buffer broadcast of DDP only add launches, so:
  * parameters are broadcast once from rank 0 (flat, coalesced);
  * zero_grad() drops the gradients (grad = None): backward then hands each parameter its gradient tensor
    without one accumulate-kernel per parameter (35 launches per step for CGCNN);
  * reduce_grads() packs them into ONE flat buffer with a single multi-tensor copy, issues a single
    all_reduce(SUM), scales by 1/world_size (DDP averages) and leaves every .grad a VIEW into that buffer;
    on one GPU it does nothing at all;
  * BatchNorm running statistics stay rank-local (as in the reference: no SyncBatchNorm) and
    rank 0's are the ones saved/evaluated.
"""
import os

import torch
import torch.distributed as dist


def ddp_setup(rank, world_size, backend=None, master_addr="127.0.0.1", master_port="12355"):
    """training.py:227-237 — env rendezvous; backend nccl (= RCCL on ROCm) on GPUs, gloo on CPU."""
    if rank in ("cpu", "cuda") or world_size <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", master_addr)
    os.environ.setdefault("MASTER_PORT", str(master_port))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        # one process per GPU: bind it before anything allocates or launches (the HIP ops launch on the current
        # device's stream) and hand the binding to RCCL so the communicator is created on this device
        local = int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=int(rank), world_size=int(world_size), **kw)
    return True


def ddp_cleanup():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


class FlatDataParallel:
    """Wraps a model in place: flat gradient storage + single all-reduce.  Not an nn.Module wrapper —
    the model keeps its own class/attributes, so `model(data)`, `state_dict()` keys and hooks are
    unchanged (the reference unwraps `model.module`; here there is nothing to unwrap)."""

    def __init__(self, model, process_group=None, broadcast=True, force=None, chunk_bytes=4 << 20):
        self.model = model
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force: run the exchange even at world size 1 (an initialised process group with one rank) — how the RCCL path is
        # exercised on a one-GPU box (MDL_FORCE_DIST=1: bench.py, tests)
        if force is None:
            force = os.environ.get("MDL_FORCE_DIST", "0") == "1"
        self.force = bool(force) and dist.is_initialized()
        self.active = self.world_size > 1 or self.force
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatDataParallel: model has no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise ValueError("FlatDataParallel expects fp32 master parameters")
            self.views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()
        # Two-chunk exchange for large payloads (MPNN: 17.5 MB): the parameters are split where the flat buffer reaches half its
        # size; the SECOND half (the layers the forward runs last) has its gradients first in the backward, so its all-reduce
        # starts from a hook as soon as the last of them is written and runs under the rest of the backward.  Small payloads
        # (CGCNN: 0.45 MB, latency bound) keep the single collective.
        self.split, self._late_left, self._late_work = None, 0, None
        if self.active and chunk_bytes and total * 4 > chunk_bytes and len(self.params) > 1:
            acc, k = 0, 0
            while k < len(self.params) - 1 and acc + self.params[k].numel() <= total // 2:
                acc += self.params[k].numel()
                k += 1
            k = max(k, 1)
            self.split = (k, sum(p.numel() for p in self.params[:k]))
            for p in self.params[k:]:
                p.register_post_accumulate_grad_hook(self._late_hook)
        self.zero_grad()
        if broadcast and self.world_size > 1:
            self.broadcast_state()

    def broadcast_state(self, src=0):
        """Parameters and buffers from rank `src`, one flat message per dtype."""
        with torch.no_grad():
            tensors = [p.data for p in self.model.parameters()] + [b.data for b in self.model.buffers()]
            by_dtype = {}
            for t in tensors:
                by_dtype.setdefault(t.dtype, []).append(t)
            for dt, ts in by_dtype.items():
                flat = torch.cat([t.reshape(-1) for t in ts])
                dist.broadcast(flat, src=src, group=self.group)
                off = 0
                for t in ts:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        if self.split is not None:
            self._late_left = len(self.params) - self.split[0]

    def single_collective(self):
        """One all-reduce per step, always (training.GraphedStep: a rank that replays a captured step and a rank that runs the
        same step eagerly must issue the same sequence of collectives)."""
        self.split = None
        return self

    def _late_hook(self, _param):
        if self.split is None:
            return
        self._late_left -= 1
        if self._late_left == 0 and self._late_work is None and self.active:
            # (not inside a stream capture: a captured step issues its collectives outside the graph)
            if self.flat_grad.is_cuda and torch.cuda.is_current_stream_capturing():
                return
            self._late_work = self._launch(self.split[0], len(self.params))

    def _launch(self, lo, hi):
        """pack the gradients of params[lo:hi] into their slice of the flat buffer and start its all_reduce(SUM)"""
        off0 = sum(p.numel() for p in self.params[:lo])
        off1 = off0 + sum(p.numel() for p in self.params[lo:hi])
        buf = self.flat_grad[off0:off1]
        if self.flat_grad.is_cuda:
            dev = self.flat_grad.device
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=dev)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                self._pack(lo, hi)
                return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pack(lo, hi)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _pack(self, lo=0, hi=None):
        srcs, dsts = [], []
        hi = len(self.params) if hi is None else hi
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:
                v.zero_()                                   # parameter unused in this step
            elif p.grad.data_ptr() != v.data_ptr():
                srcs.append(p.grad)
                dsts.append(v)
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def reduce_grads_async(self, force=False):
        """Start the gradient exchange: pack + ONE all_reduce(SUM).  On HIP devices both run on a side stream that
        waits for the backward through an event, so whatever the caller enqueues next on the compute stream (the next
        batch's assembly and RBF expansion in bench.py / the training loop) overlaps with the collective; finish() makes
        the compute stream wait for it.  Returns False when there is nothing to exchange (one rank)."""
        if not (self.active or (force and dist.is_initialized())):
            return False
        lo = 0
        if self._late_work is not None:
            lo = self.split[0]            # the late half is already on its way (hook)
        self._work = self._launch(0, lo if lo else len(self.params))
        return True

    def finish(self):
        """Wait for the exchange started by reduce_grads_async (the compute stream waits; the host does not), average
        (DDP semantics) and leave every .grad a view into the flat buffer."""
        work = getattr(self, "_work", None)
        if work is None:
            return
        work.wait()
        self._work = None
        if self._late_work is not None:
            self._late_work.wait()
            self._late_work = None
        if self.flat_grad.is_cuda:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._side)
        self.flat_grad.mul_(1.0 / self.world_size)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def reduce_grads(self, force=False):
        """Sum over ranks, then average (DDP semantics).  One pack + one collective per step."""
        if self.reduce_grads_async(force):
            self.finish()

    def grad_bytes(self):
        return self.flat_grad.numel() * 4
