"""Skeleton shared by the five reference model wrappers (each reference file repeats it:
e.g. /root/reference/matdeeplearn/models/cgcnn.py:35-119,121-174):

    pre_lin_list (+act) -> gc_count x [conv ... bn_list.i ... dropout] -> pool -> post_lin_list (+act) -> lin_out

Constructor keywords (string booleans "True"/"False", unknown keys swallowed by **kwargs), the
forward(batch) contract (`out.view(-1)` when the output dimension is 1) and the state_dict key names
are the reference's.  Extra keyword: compute_dtype = "fp32" (parity mode) | "bf16x3" (fp32 storage, split-bf16 products) | "bf16".
Dense layers are library GEMMs with fp32 master weights; everything indexed by edge_index / batch
runs on the HIP kernels of libmdl_hip.so (matdeeplearn_amd.ops / matdeeplearn_amd.nn).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..nn import BatchNorm1d, Set2Set, lowp_copies as _lowp


def dense(lin, h, split=False):
    """nn.Linear in the dtype of h with fp32 master weights.  split (the bf16x3 mode): the weight gradient of a tall fp32 layer
    on split bf16 operands (ops._LinearSplitTN)."""
    if h.dtype == lin.weight.dtype:
        if split and ops.linear_split_ok(h, lin.weight):
            return ops._LinearSplitTN.apply(h, lin.weight, lin.bias)
        return lin(h)
    return ops.linear(h, lin.weight, lin.bias, _lowp(lin))      # bf16, many rows: HIP TN GEMM for the weight gradient


def dense_act(lin, h, act, split=False):
    """getattr(F, act)(lin(h)); for bf16 activations with many rows the forward is one fused HIP kernel."""
    if h.dtype == lin.weight.dtype:
        return getattr(F, act)(dense(lin, h, split))
    return ops.linear_act(h, lin.weight, lin.bias, act, _lowp(lin))


class GraphModel(nn.Module):
    """Layer CREATION order and ModuleList REGISTRATION order are the reference's (cgcnn.py:64-119, mpnn.py:66-128,
    megnet.py:200-290): pre_lin_list, then per conv layer (conv [, gru], bn), then post_lin_list, lin_out, set2set
    [, lin_out_2].  A seeded construction therefore draws the reference's initial weights and state_dict() lists the
    reference's keys in the reference's order (tests/golden/wrappers.npz)."""

    def _begin(self, data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
               dropout_rate, compute_dtype, lists=("conv_list", "bn_list")):
        assert gc_count > 0, "Need at least 1 GC layer"
        self.batch_track_stats = batch_track_stats != "False"
        self.batch_norm, self.pool, self.act = batch_norm, pool, act
        self.pool_order, self.dropout_rate = pool_order, dropout_rate
        # "bf16x3" (round 6): fp32 storage everywhere; the CGConv kernels form their K = 2C + G product as three bf16 MFMAs on
        # (hi, lo)-split operands (ops.cgconv(split=True), MDL_SPLIT_BF16) — 16-bit operands instead of bf16's 8, 1/5 of the
        # matrix-core time of exact fp32.  Blocks without a split form (SchNet / MEGNet / MPNN / GCN) run exact fp32 under it.
        self.compute_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "bf16x3": torch.float32}[compute_dtype]
        self.split_products = compute_dtype == "bf16x3"
        self.gc_dim = data.num_features if pre_fc_count == 0 else dim1
        y0 = data[0].y
        self.output_dim = 1 if y0.ndim == 0 else len(y0[0])
        self.pre_lin_list = nn.ModuleList(
            [nn.Linear(data.num_features if i == 0 else dim1, dim1) for i in range(pre_fc_count)])
        for name in lists:
            setattr(self, name, nn.ModuleList())

    def _add_bn(self, dim):
        if self.batch_norm == "True":
            self.bn_list.append(BatchNorm1d(dim, track_running_stats=self.batch_track_stats))

    def _finish(self, dim2, post_fc_count, post_in, early_mult=1, set2set_names=("set2set",)):
        s2s_early = self.pool == "set2set" and self.pool_order == "early"
        if self.pool_order == "early":
            first_in = post_in * ((2 * early_mult - 1) if (s2s_early and early_mult > 1) else (2 if s2s_early else early_mult))
        else:
            first_in = post_in
        self.post_lin_list = nn.ModuleList(
            [nn.Linear(first_in if i == 0 else dim2, dim2) for i in range(post_fc_count)])
        self.lin_out = nn.Linear(dim2 if post_fc_count > 0 else first_in, self.output_dim)
        if s2s_early:
            for name in set2set_names:
                setattr(self, name, Set2Set(post_in, processing_steps=3))
        elif self.pool == "set2set" and self.pool_order == "late":
            setattr(self, set2set_names[0], Set2Set(self.output_dim, processing_steps=3, num_layers=1))
            self.lin_out_2 = nn.Linear(self.output_dim * 2, self.output_dim)
        if self.split_products:
            # bf16x3: the weight gradients of every Linear on split operands (the last step of every model's construction)
            from ..nn import use_split_linears
            use_split_linears(self)

    # ---- shared forward pieces ----------------------------------------------------------------
    def _inputs(self, data):
        cd = self.compute_dtype
        n = data.x.shape[0]
        csr = getattr(data, "csr", None)
        if csr is None:
            csr = ops.csr_for(data.edge_index, n)
        return data.x.to(cd), data.edge_attr.to(cd), csr

    def _cast_dense(self, dtype):
        """fp32 master weights of every dense layer -> compute dtype in ONE multi-tensor copy per forward (instead of two
        small launches per layer); the copies carry the parameters' version counters so stale ones are never used."""
        lins = getattr(self, "_dense_layers", None)           # every nn.Linear of the model, conv blocks included
        if lins is None:
            lins = [m for m in self.modules() if isinstance(m, nn.Linear)]
            object.__setattr__(self, "_dense_layers", lins)
        params = [p for lin in lins for p in (lin.weight, lin.bias) if p is not None]
        if not params or not params[0].is_cuda or params[0].dtype == dtype:
            return
        cache = getattr(self, "_lowp_cache", None)
        if cache is None or len(cache) != len(params) or cache[0].dtype != dtype or cache[0].device != params[0].device:
            cache = [torch.empty_like(p, dtype=dtype) for p in params]
            self._lowp_cache = cache
        with torch.no_grad():
            torch._foreach_copy_(cache, [p.detach() for p in params])
        k = 0
        for lin in lins:
            w = cache[k]; k += 1
            b = None
            if lin.bias is not None:
                b = cache[k]; k += 1
            lin._mdl_lowp = (w, b, lin.weight._version, None if lin.bias is None else lin.bias._version)

    def _open_grad_arena(self, device):
        """Zero-filled home of this step's dense weight gradients and BatchNorm backward sums (ops._zeros_grad): one fill
        instead of one per layer.  Opened by EVERY forward that may be followed by a backward, whatever the compute dtype — a
        stale one (left by another model or an earlier step) would hand out slices that already hold sums."""
        n = getattr(self, "_arena_floats", None)
        if n is None:
            R = ops.bn_sums_rows()
            n = sum(p.numel() + 128 for m in self.modules() if isinstance(m, nn.Linear) for p in (m.weight, m.bias) if p is not None)
            n += sum(R * m.num_features + 128 for m in self.modules() if isinstance(m, (BatchNorm1d, nn.BatchNorm1d)))
            object.__setattr__(self, "_arena_floats", n)
        ops.new_grad_arena(device, n)

    def _pre(self, out):
        if torch.is_grad_enabled() and out.is_cuda:
            self._open_grad_arena(out.device)
        if out.dtype == torch.bfloat16 and torch.is_grad_enabled():
            self._cast_dense(out.dtype)
        for lin in self.pre_lin_list:
            out = dense_act(lin, out, self.act, split=self.split_products)
        return out

    def _bn(self, i, out):
        return self.bn_list[i](out) if self.batch_norm == "True" else out

    def _drop(self, out):
        return F.dropout(out, p=self.dropout_rate, training=self.training)

    def _post(self, out, final=False):
        """final: the head's output is the model's prediction (early pooling) — the fused head then writes it as the fp32 tensor
        `out.float()` would make of its bf16 output (no cast launch forward, none in front of its backward)."""
        lins = list(self.post_lin_list) + [self.lin_out]
        if out.dtype != self.lin_out.weight.dtype and ops.mlp_head_ok(out, lins, self.act):
            # the whole head in one launch per direction (csrc/mlp.hip): separately its ~20 launches of a few microseconds
            # each are launch-bound on the pooled rows
            return ops.mlp_head(out, lins, [_lowp(lin) for lin in lins], f32_out=final)
        for lin in self.post_lin_list:
            out = dense_act(lin, out, self.act, split=self.split_products)
        return dense(self.lin_out, out)

    def _pool(self, out, data):
        num_graphs = getattr(data, "num_graphs", None)
        if self.pool == "set2set":
            return self.set2set(out, data.batch, num_graphs)
        return ops.POOLS[self.pool](out, data.batch, num_graphs, seg_index=getattr(data, "pool_index", None))

    def _head(self, out, data):
        if self.pool_order == "early":
            out = self._post(self._pool(out, data).to(out.dtype), final=True)
        else:
            out = self._post(out)
            if self.pool == "set2set":
                out = dense(self.lin_out_2, self._pool(out, data).to(out.dtype))
            else:
                out = self._pool(out, data)
        out = out.float()
        return out.view(-1) if out.shape[1] == 1 else out
