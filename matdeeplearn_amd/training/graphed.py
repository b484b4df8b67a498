"""HIP-graph replay of a whole training step ("hard parts" item 1: tiny default batches).

At the reference's batch size (config.yml:136: 100 graphs = ~32 k edges) a step is ~100 kernel launches that each run
for microseconds: Python, the autograd engine and the launch path cost 2-3 ms per step while the device needs a few hundred
microseconds — and at the large bench batch the host (5-9 ms of enqueueing once the queue is deep) falls behind the device
(3.9 ms) too.  GraphedStep captures  batch assembly (K8 + padding + K1) -> forward -> loss -> backward [-> fused AdamW]
once, on static padded buffers (process.StaticBatch), and replays it with one hipGraphLaunch per step; the host's share
drops to the upload of B graph ids.

What makes a step capturable here:
  * shapes are frozen at (n_cap, e_cap) with padding nodes that have no edges and sit in a dummy graph B; the one
    row-count-dependent operator (BatchNorm) reads the number of existing rows on the device (ops.true_rows);
  * every index structure is rebuilt inside the graph (ops.NO_INDEX_CACHE) and every scratch comes from static arenas;
  * the library never allocates or synchronises, so its launches land in the capturing stream like torch's own.
A batch that does not fit the static capacity (or a ragged last batch) runs eagerly — same arithmetic, no padding.

The padded rows (nodes past n_dev, edge slots past e_dev, the dummy graph) obey one invariant: forward values are finite,
gradients are EXACTLY zero.  Segment reductions leave them outside every segment (their backward writes zeros there),
BatchNorm over node / edge / graph rows reads the matching device row count (ops.true_rows takes a map keyed by the padded
row count), the unused edge slots point at the first padding node, and the by-source index comes from the loader
(mdl_assemble_transposed) instead of a sort of the padded arrays — so per-row dense kernels (Linear layers, weight-gradient
GEMMs over all e_cap rows) may run over the padding and add nothing.  With that, SchNet, MEGNet and GCN replay like CGCNN.
"""
import torch
import torch.nn.functional as F

from .. import ops
from ..nn import BatchNorm1d
from ..process import StaticBatch, static_capacity


class GraphedStep:
    def __init__(self, dataset, model, optimizer, batch_size, compute_dtype=torch.float32, loss="l1_loss", indices=None,
                 dp=None, capacity=None, optimizer_in_graph=None, warmup_ids=None):
        self.ds, self.model, self.opt, self.loss_name, self.dp = dataset, model, optimizer, loss, dp
        self.B, self.cdt = int(batch_size), compute_dtype
        # mean + 3.5 sigma: one batch in a few thousand takes the eager path (same arithmetic, ~2.5x the time at the reference's
        # batch size) — against 6 sigma that is 2-3 % less padded work per replay at 8192 graphs and a third less at 100
        n_cap, e_cap = capacity if capacity is not None else static_capacity(dataset, batch_size, indices, slack=3.5)
        self.sb = StaticBatch(dataset, batch_size, n_cap, e_cap, x_dtype=compute_dtype, edge_dtype=compute_dtype,
                              by_source=getattr(model, "needs_by_source", True))
        self.dev = dataset.device
        distributed = dp is not None and (dp.world_size > 1 or getattr(dp, "active", False))
        if dp is not None and hasattr(dp, "single_collective"):
            dp.single_collective()
        self.opt_in_graph = (not distributed) if optimizer_in_graph is None else bool(optimizer_in_graph)
        if self.opt_in_graph and not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ops.MdlError("GraphedStep: the optimizer step is captured — build it with capturable=True "
                               "(training.make_optimizer(..., capturable=True))")
        if self.opt_in_graph and not all(torch.is_tensor(g["lr"]) and g["lr"].is_cuda for g in optimizer.param_groups):
            raise ops.MdlError("GraphedStep: the optimizer step is captured, so its learning rate must be a device tensor (a "
                               "float would be frozen into the graph and every scheduler update ignored by the replays): "
                               "training.make_optimizer(..., capturable=True) builds it that way")
        # [loss | d loss / d prediction] of the fused loss kernel: the step's loss is read from its first element
        self._loss_buf = torch.zeros(2 + self.B + 1, dtype=torch.float32, device=self.dev)
        self.loss_value = self._loss_buf[0]
        self.graph = None
        self.bn_layers = [m for m in model.modules() if isinstance(m, BatchNorm1d)]
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.replays = self.eager_steps = 0
        self._warmup_ids = warmup_ids

    # ---- the step on the static buffers (what gets captured) ------------------------------------------------------
    def _body(self):
        sb = self.sb
        batch = sb.assemble()
        with ops.true_rows(batch.true_rows), ops.zero_arena(self.dev):
            out = self.model(batch)
            # the dummy graph's prediction (row B) is outside the loss: its gradient is written as zero by the loss kernel itself,
            # and the root gradient is the constant the loss node hands through (no slice / fill / multiply launches)
            loss = (ops.loss(self.loss_name, out, sb.y, rows=self.B, buf=self._loss_buf) if out.dim() == 1
                    else ops.loss(self.loss_name, out[:self.B], sb.y))
            ops.backward(loss)
        if loss.data_ptr() != self.loss_value.data_ptr():
            self.loss_value.copy_(loss.detach())
        if self.opt_in_graph:
            self.opt.step()

    def capture(self, ids):
        """Warm up on a side stream (allocator / lazy-init effects must not land in the graph), then capture."""
        if not self.sb.fits(ids):
            raise ops.MdlError("GraphedStep.capture: the warm-up batch does not fit the static capacity")
        self.model.train()
        prev = ops.NO_INDEX_CACHE
        ops.NO_INDEX_CACHE = True
        try:
            self.sb.load(ids)
            # The warm-up iterations run the real step (allocator pools, lazy optimizer state, low-precision weight copies
            # must exist before the capture) — but they must not train: parameters, buffers, optimizer state and the
            # host-side BatchNorm step counters are put back afterwards, so the captured step is the FIRST step.
            keep = [t for t in list(self.model.parameters()) + list(self.model.buffers())]
            saved = [t.detach().clone() for t in keep]
            fresh_opt = len(self.opt.state) == 0
            opt_saved = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                         for p, st in self.opt.state.items()}
            nbt = [getattr(m, "_nbt_pending", 0) for m in self.bn_layers]
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._zero_grad()
                    self._body()
                    if not self.opt_in_graph:
                        # no gradient exchange in the warm-up: its results are thrown away, and a rank whose first batch
                        # takes the eager path instead issues ONE all-reduce for this step — the capture must issue exactly
                        # one too (the replay below), or equal-sized collectives of different steps pair up across ranks
                        self._finish_eager(comm=False)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            self._zero_grad()
            self.graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: with a process group alive, its watchdog thread polls the events of earlier collectives
            # (hipEventQuery) at its own pace — under the default GLOBAL mode such a call from ANOTHER thread during the capture is
            # "operation not permitted when stream is capturing" and takes the process down (seen once in five runs of the RCCL
            # world-1 test, round 6); only this thread's own calls belong to the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self._body()
            self.static_grads = [p.grad for p in self.params]
            with torch.no_grad():
                for t, s in zip(keep, saved):
                    t.copy_(s)
                for p, st in self.opt.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if fresh_opt:
                                v.zero_()
                            else:
                                v.copy_(opt_saved[id(p)][k])
            for m, n in zip(self.bn_layers, nbt):
                m._nbt_pending = n
            # the capture itself executed nothing; replay once so that this batch's step is actually taken
            self._replay()
        finally:
            ops.NO_INDEX_CACHE = prev
        return self

    def _zero_grad(self):
        for p in self.params:
            p.grad = None

    def _bump_bn(self, n=1):
        for m in self.bn_layers:                                   # num_batches_tracked is counted on the host
            if m.training and m.track_running_stats:
                m._nbt_pending = getattr(m, "_nbt_pending", 0) + n

    def _finish_eager(self, comm=True):
        if self.dp is not None and comm:
            self.dp.reduce_grads()
        self.opt.step()

    def _stale_lowp(self):
        """The captured optimizer step rewrites the parameters without Python noticing (no version counter moves on a replay), so
        the per-layer bf16 copies that the captured forward made at the START of the step would still pass for current: drop the
        record — an eager forward behind a replay (validation, the final evaluation of a job) casts the masters itself."""
        for lin in getattr(self.model, "_dense_layers", None) or ():
            lin._mdl_lowp = None

    def _replay(self):
        self.graph.replay()
        self._bump_bn(1)
        self.replays += 1
        if self.opt_in_graph:
            self._stale_lowp()
        if not self.opt_in_graph:
            for p, g in zip(self.params, self.static_grads):       # the graph writes into ITS gradient tensors
                p.grad = g
            self._finish_eager()

    # ---- forward-only evaluation on the same static buffers (validation between the epochs) ------------------------------
    def _eval_body(self):
        sb = self.sb
        batch = sb.assemble()
        with torch.no_grad(), ops.true_rows(batch.true_rows), ops.zero_arena(self.dev):
            # (the bf16 copies of the dense weights are refreshed by the TRAINING forward, i.e. they predate the last optimizer step:
            # the evaluation casts the current masters itself)
            if self.cdt == torch.bfloat16 and hasattr(self.model, "_cast_dense"):
                self.model._cast_dense(self.cdt)
            out = self.model(batch)
            if out.dim() == 1 and out.dtype == torch.float32:
                ops.loss(self.loss_name, out, sb.y, rows=self.B, buf=self._eval_buf)
            else:
                self._eval_buf[0].copy_(getattr(F, self.loss_name)(out[:self.B].float(), sb.y.view_as(out[:self.B])))

    def capture_eval(self, ids):
        """Capture the forward-only step (eval mode: BatchNorm on its running statistics, no dropout)."""
        self._eval_buf = torch.zeros(2 + self.B + 1, dtype=torch.float32, device=self.dev)
        was = self.model.training
        self.model.eval()
        for m in self.bn_layers:                                   # (the host-side step counters reach their buffers OUTSIDE the capture)
            m._sync_counter()
        prev = ops.NO_INDEX_CACHE
        ops.NO_INDEX_CACHE = True
        try:
            self.sb.load(ids)
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._eval_body()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            self.eval_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.eval_graph, capture_error_mode="thread_local"):
                self._eval_body()
        finally:
            ops.NO_INDEX_CACHE = prev
            self.model.train(was)
        return self

    def eval_loss(self, ids):
        """Mean loss of the graphs `ids` under the current weights in eval mode, as a device scalar that stays valid until the next
        call (a view of the evaluation's persistent buffer): ONE replay for a full batch that fits the static capacity, the eager
        forward otherwise.  The model is left in the mode it was in."""
        if self.sb.fits(ids):
            if getattr(self, "eval_graph", None) is None:
                self.capture_eval(ids)
            for m in self.bn_layers:
                m._sync_counter()
            self.sb.load(ids)
            self.eval_graph.replay()
            self.eval_replays = getattr(self, "eval_replays", 0) + 1
            return self._eval_buf[0]
        was = self.model.training
        self.model.eval()
        try:
            with torch.no_grad():
                batch = self.ds.collate(ids, edge_dtype=self.cdt, x_dtype=self.cdt)
                out = self.model(batch)
                return getattr(F, self.loss_name)(out, batch.y.view_as(out))
        finally:
            self.model.train(was)

    # ---- public step ------------------------------------------------------------------------------------------------
    def step(self, ids):
        """One training step on the graphs `ids`.  Returns (edges, nodes) of the batch (true counts, no padding)."""
        if self.sb.fits(ids):
            if self.graph is None:
                self.capture(ids)
            else:
                self.sb.load(ids)
                self._replay()
            return self.sb.true_edges, self.sb.true_nodes
        return self._eager(ids)

    def _eager(self, ids):
        batch = self.ds.collate(ids, edge_dtype=self.cdt, x_dtype=self.cdt)
        self._zero_grad()
        with ops.zero_arena(self.dev):
            out = self.model(batch)
            loss = ops.loss(self.loss_name, out, batch.y, buf=self._loss_buf)
            ops.backward(loss)
        if loss.data_ptr() != self.loss_value.data_ptr():
            self.loss_value.copy_(loss.detach())
        self._finish_eager()
        self.eager_steps += 1
        return batch.num_edges, batch.num_nodes
