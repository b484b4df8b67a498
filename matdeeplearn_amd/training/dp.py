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
    unchanged (the reference unwraps `model.module`; here there is nothing to unwrap).

    chunk_bytes (default 0 = off): payloads above it are exchanged in TWO chunks.  The chunk membership is not guessed from
    `parameters()` (registration order is not backward order: GraphModel registers every bn_list.* after every conv_list.*)
    but OBSERVED: the first step runs the single collective while gradient hooks record the order in which the gradients
    become ready; rank 0's order is broadcast (one agreed layout), the flat buffer is re-laid out in that order, and from the
    second step on the first-ready half is packed and all-reduced from the hook of its last member, under the rest of the
    backward.  Every step then issues exactly two collectives in the same order on every rank — a parameter that got no
    gradient on this rank only delays its chunk to reduce_grads(), where it is packed as zeros.  Gradient accumulation (two
    backward passes before one exchange) is supported: the early chunk that left after the first backward is re-packed from
    the accumulated gradients and reduced again by reduce_grads_async (three collectives in that step, on every rank)."""

    def __init__(self, model, process_group=None, broadcast=True, force=False, chunk_bytes=0):
        self.model = model
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force: run the exchange even at world size 1 (an initialised process group with one rank) — how the RCCL path is
        # exercised on a one-GPU box (bench.py --force-dist, tests); no environment variable is read
        self.force = bool(force) and dist.is_initialized()
        self.active = self.world_size > 1 or self.force
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatDataParallel: model has no trainable parameters")
        for p in self.params:
            if p.dtype != torch.float32:
                raise ValueError("FlatDataParallel expects fp32 master parameters")
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self._layout(self.params)
        self._work = self._early_work = None
        self._side = None
        # two-chunk exchange: None = off; "observe" = the next backward records the ready order; (k, numel) = params[:k]
        # (the first `numel` floats of the flat buffer) are the chunk whose exchange starts from a hook
        self.split = None
        self._ready, self._early_left = [], 0
        self._early_stale = False
        if self.active and chunk_bytes and total * 4 > chunk_bytes and len(self.params) > 1:
            self.split = "observe"
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        self.zero_grad()
        if broadcast and self.world_size > 1:
            self.broadcast_state()

    def _layout(self, params):
        """views of the flat buffer in the order of `params`"""
        self.params, self.views, off = list(params), [], 0
        for p in self.params:
            self.views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += p.numel()

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

    def broadcast_buffers(self, src=0):
        """The model's buffers (BatchNorm running statistics, which stay rank-local during training) from rank `src`, one flat
        message per dtype: what DDP's broadcast_buffers=True does before EVERY forward (the reference's default,
        training.py:263-266) is needed here only where ranks must evaluate the same model — the sharded validation."""
        if self.world_size <= 1:
            return
        with torch.no_grad():
            by_dtype = {}
            for b in self.model.buffers():
                by_dtype.setdefault(b.dtype, []).append(b.data)
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
        self._ready = []
        if self._early_work is not None:                  # a backward whose gradients were never reduced: drop its exchange
            self._early_work.wait()
            self._early_work = None
        self._early_stale = False
        if isinstance(self.split, tuple):
            self._early_left = self.split[0]

    def single_collective(self):
        """One all-reduce per step, always (training.GraphedStep: a rank that replays a captured step and a rank that runs the
        same step eagerly must issue the same sequence of collectives)."""
        self.split = None
        return self

    def _hook(self, param):
        if self.split is None:
            return
        if self.split == "observe":
            self._ready.append(param)
            return
        if id(param) not in self._early_ids:
            return
        self._early_left -= 1
        if self._early_left < 0:
            # a SECOND backward before the exchange (gradient accumulation): the early chunk left after the first one with that
            # micro-batch's gradients only — reduce_grads_async re-packs and re-reduces it from the accumulated .grad tensors
            self._early_stale = True
        if self._early_left == 0 and self._early_work is None:
            # (not inside a stream capture: a captured step issues its collectives outside the graph)
            if self.flat_grad.is_cuda and torch.cuda.is_current_stream_capturing():
                return
            self._early_work = self._launch(0, self.split[0])

    def _adopt_ready_order(self):
        """End of the observed step: agree on rank 0's ready order, lay the flat buffer out in it, split it in two."""
        # (two backward passes before one exchange — gradient accumulation, a warm-up backward without dp.zero_grad() — fire
        # every hook twice: keep each parameter's FIRST appearance)
        seen, ready = set(), []
        for p in self._ready:
            if id(p) not in seen:
                seen.add(id(p))
                ready.append(p)
        order = ready + [p for p in self.params if id(p) not in seen]              # never-ready parameters go last
        assert len(order) == len(self.params), "ready order must be a permutation of the parameters"
        index = {id(p): i for i, p in enumerate(self.params)}
        perm = torch.tensor([index[id(p)] for p in order], dtype=torch.int64, device=self.flat_grad.device)
        if self.world_size > 1:
            dist.broadcast(perm, src=0, group=self.group)
        old = self.params
        new = [old[i] for i in perm.tolist()]
        # the gradients of this step already sit in the flat buffer in the OLD layout: move them (one gather, off the hot path)
        moved = torch.cat([v.reshape(-1) for v in (self.views[i] for i in perm.tolist())])
        self.flat_grad.copy_(moved)
        self._layout(new)
        total, acc, k = self.flat_grad.numel(), 0, 0
        while k < len(new) - 1 and acc + new[k].numel() <= total // 2:
            acc += new[k].numel()
            k += 1
        k = max(k, 1)
        self.split = (k, sum(p.numel() for p in new[:k]))
        self._early_ids = {id(p) for p in new[:k]}

    def _launch(self, lo, hi):
        """pack the gradients of params[lo:hi] into their slice of the flat buffer and start its all_reduce(SUM)"""
        off0 = sum(p.numel() for p in self.params[:lo])
        off1 = off0 + sum(p.numel() for p in self.params[lo:hi])
        buf = self.flat_grad[off0:off1]
        if self.flat_grad.is_cuda:
            dev = self.flat_grad.device
            if self._side is None:
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
        """Start the gradient exchange: pack + all_reduce(SUM) (one, or the second of two when the first-ready chunk left
        from its hook).  On HIP devices both run on a side stream that waits for the backward through an event, so whatever
        the caller enqueues next on the compute stream (the next batch's assembly and RBF expansion in bench.py / the
        training loop) overlaps with the collective; finish() makes the compute stream wait for it.  Returns False when
        there is nothing to exchange (one rank)."""
        if not (self.active or (force and dist.is_initialized())):
            return False
        lo = 0
        if isinstance(self.split, tuple):
            if self._early_work is not None and self._early_stale:
                self._early_work.wait()                   # (accumulation: every rank fires its hooks twice, so every rank is here)
                self._early_work = None
            if self._early_work is None:                  # a member got no gradient on this rank: same two collectives, now
                self._early_work = self._launch(0, self.split[0])
            self._early_stale = False
            lo = self.split[0]
        self._work = self._launch(lo, len(self.params))
        return True

    def finish(self):
        """Wait for the exchange started by reduce_grads_async (the compute stream waits; the host does not), average
        (DDP semantics) and leave every .grad a view into the flat buffer."""
        work = self._work
        if work is None:
            return
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None
        work.wait()
        self._work = None
        if self.flat_grad.is_cuda:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._side)
        if self.split == "observe":
            self._adopt_ready_order()
        self.flat_grad.mul_(1.0 / self.world_size)
        for p, v in zip(self.params, self.views):
            p.grad = v
        # the exchange is over: the next backward counts its early chunk from the start, whoever clears the gradients
        # (dp.zero_grad() or the optimizer's zero_grad())
        self._ready = []
        self._early_stale = False
        if isinstance(self.split, tuple):
            self._early_left = self.split[0]

    def reduce_grads(self, force=False):
        """Sum over ranks, then average (DDP semantics).  One pack + one collective per step (two with chunk_bytes)."""
        if self.reduce_grads_async(force):
            self.finish()

    def grad_bytes(self):
        return self.flat_grad.numel() * 4
