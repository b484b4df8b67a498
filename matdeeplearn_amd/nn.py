"""Convolution operator modules with the PyG constructor / forward surface the reference uses
(SURVEY.md 8b level b2), backed by the HIP kernels in libmdl_hip.so.

  CGConv(channels, dim, aggr, batch_norm)   matdeeplearn/models/cgcnn.py:80-83 / :136-145
Sub-module and parameter names follow PyG so `state_dict()` keys interchange with the reference
(SURVEY Appendix A.7): CGConv has `lin_f`, `lin_s` = Linear(2*channels + dim, channels).
"""
from torch import nn

from . import ops


class CGConv(nn.Module):
    """out_i = x_i + aggr_{j->i} sigmoid(lin_f z_ij) * softplus(lin_s z_ij), z_ij = [x_i | x_j | e_ij]."""

    def __init__(self, channels, dim=0, aggr="add", batch_norm=False, bias=True, **kwargs):
        super().__init__()
        if isinstance(channels, (tuple, list)):
            if channels[0] != channels[1]:
                raise ops.MdlError("CGConv: bipartite channel pairs are not on the reference path")
            channels = channels[0]
        if batch_norm:
            raise ops.MdlError("CGConv(batch_norm=True) is not used by the reference (cgcnn.py:81)")
        self.channels, self.dim, self.aggr = channels, dim, aggr
        # same construction order / default init as PyG (torch.nn.Linear kaiming-uniform)
        self.lin_f = nn.Linear(2 * channels + dim, channels, bias=bias)
        self.lin_s = nn.Linear(2 * channels + dim, channels, bias=bias)
        self.reset_parameters()          # as upstream: the constructor ends with a second draw of both layers

    def reset_parameters(self):
        self.lin_f.reset_parameters()
        self.lin_s.reset_parameters()

    def forward(self, x, edge_index, edge_attr=None, csr=None, bn=None, bn_shift=None, packed=None, split=False):
        """PyG's forward(x, edge_index, edge_attr).  Extra keywords of this build: `csr` (the batch's ops.EdgeCSR; no lookup by
        edge_index), and `bn` — the BatchNorm1d the caller applies to the result (cgcnn.py:143): when it normalises with batch
        statistics and the layer runs on the static bf16 kernels, bn(conv(x)) is returned with the statistics formed in the
        conv kernel's epilogue (`bn_shift`: [C] values near the column means, e.g. the beta of the BatchNorm in front);
        `packed`: this layer's entry of ops.cgconv_prepack (its weights packed with the model's other layers in one launch);
        `split`: fp32 tensors only — the kernels' products on (hi, lo)-split bf16 operands (MDL_SPLIT_BF16, "bf16x3")."""
        if edge_attr is None:
            edge_attr = x.new_zeros((edge_index.shape[1], 0))
        if bn is not None:
            if csr is None:
                csr = ops.csr_for(edge_index, x.shape[0])
            y = bn.after_cgconv(self, x, edge_index, edge_attr, csr, bn_shift, packed)
            if y is not None:
                return y
        y = ops.cgconv(x, edge_index, edge_attr, self.lin_f.weight, self.lin_f.bias, self.lin_s.weight,
                       self.lin_s.bias, self.aggr, csr=csr, packed=packed, split=split)
        return y if bn is None else bn(y)

    def extra_repr(self):
        return "%d, dim=%d, aggr=%s" % (self.channels, self.dim, self.aggr)


class SplitLinear(nn.Linear):
    """nn.Linear of a model in the bf16x3 mode (models/_base.py swaps the class of every Linear: same parameters, same
    state_dict keys): forward and dX are the library's exact fp32 products; the WEIGHT GRADIENT of a tall input — g^T x with the
    contraction over 1e5 .. 1e6 node or edge rows, which the library's fp32 path runs at ~5 TFLOP/s (half of MEGNet's exact-fp32
    step) — runs as three bf16 TN-GEMM launches on (hi, lo)-split operands (ops._LinearSplitTN)."""

    def forward(self, x):
        if x.dim() == 2 and ops.linear_split_ok(x, self.weight):
            return ops._LinearSplitTN.apply(x, self.weight, self.bias)
        return super().forward(x)


def use_split_linears(model):
    """bf16x3 mode: every plain nn.Linear of `model` becomes a SplitLinear (in place; nothing else changes)"""
    for m in model.modules():
        if type(m) is nn.Linear:
            m.__class__ = SplitLinear
    return model


# ------------------------------------------------------------------------------------------------
# SchNet: torch_geometric.nn.models.schnet.{ShiftedSoftplus, CFConv, InteractionBlock} (2.0.1) as
# constructed at matdeeplearn/models/schnet.py:81 and called at schnet.py:134-143.
# The filter MLP and the node Linears are library GEMMs; the gather * filter * cutoff -> segmented
# sum is the K4a HIP kernel (forward, and its transpose for the gradient).
# ------------------------------------------------------------------------------------------------
import math

import torch
import torch.nn.functional as F


def lowp_copies(lin):
    """The layer's (weight, bias) in the compute dtype if the model's per-step multi-tensor cast (GraphModel._cast_dense)
    made them for the CURRENT parameter values (version counters), else None."""
    sh = getattr(lin, "_mdl_lowp", None)
    if sh is None or sh[2] != lin.weight._version or (lin.bias is not None and sh[3] != lin.bias._version):
        return None
    return sh[0], sh[1]


def _lin(lin, h, act=None, in_act=None, out_pre=False, pre=None):
    """nn.Linear (+ activation) applied in the dtype of h (fp32 master weights, bf16 activations): the streaming HIP dense
    layer for bf16 rows, the library otherwise."""
    if h.dtype == lin.weight.dtype:
        y = lin(h)
        if act is None:
            return y
        return F.softplus(y) - math.log(2.0) if act == "ssp" else getattr(F, act)(y)
    return ops.linear_act(h, lin.weight, lin.bias, act, lowp_copies(lin), in_act, out_pre, pre)


def _seq(seq, h, pre=None):
    """Sequential of Linear / activation modules; a Linear followed by ShiftedSoftplus / ReLU runs as ONE fused dense layer.
    Between two fused layers of the chain the intermediate tensor is private to this function, so the activation derivative
    is handed down the chain (ops._LinearActTN: the later layer's backward returns the gradient w.r.t. the earlier layer's
    pre-activation, the earlier layer's backward reads neither its output nor applies a derivative).
    pre: the outputs of the chain's Linear layers, already formed by a fused forward (CFConv: ops.cfconv_fused) — every layer
    must then take the fused dense path, whose autograd node only records the graph."""
    mods = list(seq)
    layers, k = [], 0                                   # (module, act) per step; act is None for non-Linear modules
    while k < len(mods):
        m = mods[k]
        if isinstance(m, nn.Linear):
            a = mods[k + 1] if k + 1 < len(mods) else None
            act = "ssp" if isinstance(a, ShiftedSoftplus) else ("relu" if isinstance(a, nn.ReLU) else None)
            layers.append((m, act))
            k += 2 if act else 1
        else:
            layers.append((m, False))
            k += 1
    handed = False                                      # the previous step was a fused layer told to expect a pre-activation gradient
    pre = list(pre) if pre is not None else None
    for j, (m, act) in enumerate(layers):
        if act is False:
            assert pre is None
            h, handed = m(h), False
            continue
        nxt = layers[j + 1] if j + 1 < len(layers) else None
        fused = h.dtype != m.weight.dtype and ops.linear_act_fused_ok(h, m.weight, act)
        # hand this layer's derivative to the next one: both fused, this one activated, the next one a Linear on our output
        give = (fused and act in ("relu", "ssp") and nxt is not None and nxt[1] is not False and torch.is_grad_enabled()
                and nxt[0].weight.requires_grad and nxt[0].in_features == m.out_features
                and ops._hip_shape_ok(nxt[0].out_features, nxt[0].in_features)
                and (nxt[1] != "ssp" or nxt[0].out_features % 2 == 0) and h.shape[0] >= ops._DENSE_MIN_ROWS)
        prev_act = layers[j - 1][1] if handed else None
        if fused:
            h = _lin(m, h, act, in_act=prev_act, out_pre=give, pre=pre.pop(0) if pre else None)
        else:
            assert not handed and pre is None
            h = _lin(m, h, act)
        handed = give
    return h


class ShiftedSoftplus(nn.Module):
    def __init__(self):
        super().__init__()
        self.shift = math.log(2.0)

    def forward(self, x):
        return F.softplus(x) - self.shift


def cosine_cutoff(edge_weight, cutoff):
    """0.5 (cos(d pi / cutoff) + 1) of a batch's distances, [E] fp32: a function of the batch alone.  A model evaluates it
    ONCE per forward and hands it to its interaction blocks (`cut=`) instead of four elementwise launches per block; nothing
    is cached across calls (a cache keyed on the tensor would go stale on the static buffers of the HIP-graph path, which
    kernels rewrite in place, and would pin the previous batch)."""
    return 0.5 * (torch.cos(edge_weight.float() * (math.pi / cutoff)) + 1.0)


class CFConv(nn.Module):
    def __init__(self, in_channels, out_channels, num_filters, net, cutoff):
        super().__init__()
        self.lin1 = nn.Linear(in_channels, num_filters, bias=False)
        self.lin2 = nn.Linear(num_filters, out_channels)
        self.nn = net
        self.cutoff = cutoff
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.lin1.weight)
        nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)

    def forward(self, x, edge_index, edge_weight, edge_attr, csr=None, cut=None, by_source=None):
        return _lin(self.lin2, self.aggregate(x, edge_index, edge_weight, edge_attr, csr=csr, cut=cut, by_source=by_source))

    def aggregate(self, x, edge_index, edge_weight, edge_attr, csr=None, cut=None, by_source=None):
        """sum_j lin1(x)_j * W(e_ij) * C(d_ij): the convolution in front of lin2 (InteractionBlock chains lin2 -> ssp -> lin as
        ONE fused sequence behind it)"""
        if csr is None:
            csr = ops.csr_for(edge_index, x.shape[0])
        c = cosine_cutoff(edge_weight, self.cutoff) if cut is None else cut           # [E] fp32
        h = _lin(self.lin1, x)
        mods = list(self.nn)
        if (len(mods) == 3 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], ShiftedSoftplus) and isinstance(mods[2], nn.Linear)
                and ops.cfconv_fused_ok(edge_attr, h, csr, mods[0], mods[2])):
            if (ops._CFCONV_RECOMPUTE and h.shape[1] >= ops._CFCONV_RECOMPUTE_MIN_F and not c.requires_grad
                    and not edge_attr.requires_grad):
                # K4 + K4b: one autograd node, nothing stored per edge; the backward recomputes the filter (by_source: the edge
                # features in by-source order, shared by the blocks of a model)
                return ops.cfconv_recompute(edge_attr, c, h, csr, mods[0], mods[2], by_source)
            # K4: filter network, cutoff, h[src] * W and the segmented sum in ONE pass over the edges.  Under autograd the two
            # dense layers and the gather-multiply-reduce keep their nodes (the backward is theirs) — they receive the
            # activations the fused pass wrote instead of launching their own forward kernels.
            train = torch.is_grad_enabled() and (h.requires_grad or mods[0].weight.requires_grad)
            agg, a1, w = ops.cfconv_fused(edge_attr, c, h.detach(), csr, mods[0], mods[2], want_acts=train)
            if train:
                w = _seq(self.nn, edge_attr, pre=[a1, w])
                agg = ops.gather_mul_reduce(h, csr, w=w, scale=c, reduce="sum", pre=agg)
            return agg
        w = _seq(self.nn, edge_attr)                                                  # filter  [E, F]
        return ops.gather_mul_reduce(h, csr, w=w.to(h.dtype), scale=c, reduce="sum")


class InteractionBlock(nn.Module):
    def __init__(self, hidden_channels, num_gaussians, num_filters, cutoff):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(num_gaussians, num_filters), ShiftedSoftplus(),
                                 nn.Linear(num_filters, num_filters))
        self.conv = CFConv(hidden_channels, hidden_channels, num_filters, self.mlp, cutoff)
        self.act = ShiftedSoftplus()
        self.lin = nn.Linear(hidden_channels, hidden_channels)
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.mlp[0], self.mlp[2]):            # upstream order: filter network, conv, output layer
            nn.init.xavier_uniform_(m.weight)
            m.bias.data.fill_(0)
        self.conv.reset_parameters()
        nn.init.xavier_uniform_(self.lin.weight)
        self.lin.bias.data.fill_(0)

    def forward(self, x, edge_index, edge_weight, edge_attr, csr=None, cut=None, by_source=None):
        # lin(ssp(lin2(agg))) as one chain: fused dense layers with the activation in the first one's epilogue and its derivative
        # handed down from the second one's backward (no softplus / softplus_backward passes over [N, C])
        agg = self.conv.aggregate(x, edge_index, edge_weight, edge_attr, csr=csr, cut=cut, by_source=by_source)
        return _seq([self.conv.lin2, self.act, self.lin], agg)


# ------------------------------------------------------------------------------------------------
# GCNConv(improved=True, add_self_loops=False) with edge_weight = raw distance — gcn.py:80-82,135-144 (A.5)
# ------------------------------------------------------------------------------------------------
class GCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, add_self_loops=True, bias=True, **kwargs):
        super().__init__()
        if add_self_loops:
            raise ops.MdlError("GCNConv(add_self_loops=True) is not on the reference path (gcn.py:81)")
        # two glorot draws like PyG 2.0.1 (its Linear draws once in the ctor — without torch's default-init draw —, then
        # GCNConv.__init__ ends with reset_parameters()): a third draw would shift every later layer's seeded weights
        self.lin = torch.nn.utils.skip_init(nn.Linear, in_channels, out_channels, bias=False)
        nn.init.xavier_uniform_(self.lin.weight)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.lin.weight)        # glorot, drawn again by the upstream constructor's reset
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, edge_index, edge_weight=None, csr=None):
        if csr is None:
            csr = ops.csr_for(edge_index, x.shape[0])
        n = x.shape[0]
        if edge_weight is None:
            edge_weight = torch.ones(csr.E, device=x.device)
        ew = edge_weight.float()
        deg = ops.scatter(ew.unsqueeze(1), csr.col, 0, n, "sum").squeeze(1)            # weighted in-degree
        dis = deg.pow(-0.5)
        dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
        norm = ops.gather(dis.unsqueeze(1), csr.row).squeeze(1) * ew * ops.gather(dis.unsqueeze(1), csr.col).squeeze(1)
        out = ops.gather_mul_reduce(_lin(self.lin, x), csr, w=None, scale=norm, reduce="sum")
        return out + self.bias.to(out.dtype) if self.bias is not None else out


# ------------------------------------------------------------------------------------------------
# NNConv(aggr="mean") — mpnn.py:83-88,148-157 (A.4).  K7: with the last layer of the edge network a
# Linear(d3, C_in*C_out) the message is re-associated as m_e = Y[src_e] . h_e + Z[src_e] with
# Y = x @ W2.view(C_in, C_out*d3), Z = x @ b2.view(C_in, C_out) (two dense GEMMs over the nodes) and the
# per-edge mat-vec runs in csrc/nnconv.hip: neither the E x C^2 tensor of the reference nor per-edge
# C x C GEMMs exist (2 MFLOP -> 20 kFLOP per edge at C = d3 = 100).  Other edge networks fall back to
# chunked per-edge GEMMs.
# ------------------------------------------------------------------------------------------------
class NNConv(nn.Module):
    def __init__(self, in_channels, out_channels, nn_module, aggr="add", root_weight=True, bias=True, chunk=32768):
        super().__init__()
        self.in_channels, self.out_channels, self.aggr, self.chunk = in_channels, out_channels, aggr, chunk
        self.nn = nn_module
        self.lin = nn.Linear(in_channels, out_channels, bias=False) if root_weight else None
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.nn.modules():                      # upstream reset(self.nn), then root weight, bias = 0
            if m is not self.nn and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        if self.lin is not None:
            self.lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.zero_()

    def _messages(self, xj, edge_attr):
        w = self.nn(edge_attr).view(-1, self.in_channels, self.out_channels)
        return torch.bmm(xj.unsqueeze(1), w).squeeze(1)

    def _last_linear(self):
        """The edge network's last layer if it is the Linear(d3, C_in*C_out) the re-association needs, else None."""
        last = self.nn[-1] if isinstance(self.nn, nn.Sequential) and len(self.nn) > 0 else None
        if isinstance(last, nn.Linear) and last.out_features == self.in_channels * self.out_channels:
            return last
        return None

    def forward(self, x, edge_index, edge_attr, csr=None):
        from torch.utils.checkpoint import checkpoint
        if csr is None:
            csr = ops.csr_for(edge_index, x.shape[0])
        last = self._last_linear()
        ci, co = self.in_channels, self.out_channels
        if last is not None and (co * (last.in_features + 1) + last.in_features + co) * 4 <= 160 * 1024:
            hdn = _seq(list(self.nn)[:-1], edge_attr)
            d3 = last.in_features
            w2 = last.weight.view(ci, co * d3).to(x.dtype)
            Y = ops.matmul_wide(x, w2)                                    # [N, C_out*d3]: the only large dense product
            m = ops.nnconv_msg(Y, hdn.to(x.dtype), csr, co)
            linear_aggr = self.aggr in ("mean", "add", "sum")
            if last.bias is not None and not linear_aggr:
                m = m + ops.gather(x @ last.bias.view(ci, co).to(x.dtype), csr.row)
            out = ops.scatter(m, csr.col, 0, x.shape[0], self.aggr)
            if last.bias is not None and linear_aggr:
                # the bias of the edge network's last layer contributes x_j B2 (B2 = bias.view(C_in, C_out)) to every message: a
                # per-NODE row gathered by source.  Sum and mean distribute over it, so it is aggregated on its own — K4a reads the
                # [N, C_out] rows per edge from L2 — instead of as an [E, C_out] gather + add in front of the scatter (and a
                # by-source segment sum + an [E, C_out] gradient copy behind it): ~90 us per layer at cfg5's 8e5 edges
                out = out + ops.gather_mul_reduce(x @ last.bias.view(ci, co).to(x.dtype), csr, reduce=self.aggr)
            if self.lin is not None:
                out = out + _lin(self.lin, x)
            if self.bias is not None:
                out = out + self.bias.to(out.dtype)
            return out
        xj = ops.gather(x, csr.row)                                   # caller's edge order
        parts = []
        for s in range(0, csr.E, self.chunk):
            a, b = xj[s:s + self.chunk], edge_attr[s:s + self.chunk]
            parts.append(checkpoint(self._messages, a, b, use_reentrant=False) if torch.is_grad_enabled()
                         else self._messages(a, b))
        m = torch.cat(parts) if parts else xj.new_zeros((0, self.out_channels))
        out = ops.scatter(m, csr.col, 0, x.shape[0], self.aggr)
        if self.lin is not None:
            out = out + self.lin(x)
        if self.bias is not None:
            out = out + self.bias.to(out.dtype)
        return out


# ------------------------------------------------------------------------------------------------
# MetaLayer — megnet.py:235-253 (A.6); Set2Set — cgcnn.py:112-119 (A.6)
# ------------------------------------------------------------------------------------------------
class MetaLayer(nn.Module):
    def __init__(self, edge_model=None, node_model=None, global_model=None):
        super().__init__()
        self.edge_model, self.node_model, self.global_model = edge_model, node_model, global_model

    def forward(self, x, edge_index, edge_attr=None, u=None, batch=None):
        row, col = edge_index[0], edge_index[1]
        if self.edge_model is not None:
            edge_attr = self.edge_model(ops.gather(x, row), ops.gather(x, col), edge_attr, u,
                                        batch if batch is None else batch.index_select(0, row))
        if self.node_model is not None:
            x = self.node_model(x, edge_index, edge_attr, u, batch)
        if self.global_model is not None:
            u = self.global_model(x, edge_index, edge_attr, u, batch)
        return x, edge_attr, u


class Set2Set(nn.Module):
    def __init__(self, in_channels, processing_steps, num_layers=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = nn.LSTM(self.out_channels, in_channels, num_layers)
        self.lstm.reset_parameters()

    def forward(self, x, batch, size=None):
        b = int(batch.max()) + 1 if size is None else size
        xf = x.float()
        h = (xf.new_zeros((self.num_layers, b, self.in_channels)), xf.new_zeros((self.num_layers, b, self.in_channels)))
        q_star = xf.new_zeros(b, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self.lstm(q_star.unsqueeze(0), h)
            q = q.view(b, self.in_channels)
            e = (xf * ops.gather(q, batch)).sum(dim=-1, keepdim=True)
            emax = ops.gather(ops.scatter(e, batch, 0, b, "max", assume_sorted=True).detach(), batch)
            a = torch.exp(e - emax)
            a = a / (ops.gather(ops.scatter(a, batch, 0, b, "sum", assume_sorted=True), batch) + 1e-16)
            r = ops.scatter(a * xf, batch, 0, b, "sum", assume_sorted=True)
            q_star = torch.cat([q, r], dim=-1)
        return q_star


# ------------------------------------------------------------------------------------------------
# BatchNorm1d — same parameters / buffers / state_dict keys as torch.nn.BatchNorm1d (the reference
# constructs BatchNorm1d(gc_dim, track_running_stats=...) at cgcnn.py:85-87); training-mode forward
# and backward run on the HIP stream kernels when the shape allows, otherwise the library path.
# ------------------------------------------------------------------------------------------------
class BatchNorm1d(nn.BatchNorm1d):
    """torch.nn.BatchNorm1d whose training-mode forward / backward run on the HIP kernels.  `num_batches_tracked` is
    counted on the host and folded into the buffer when the state is read (state_dict, or the library path that uses
    it): a one-element device add per layer per step is a kernel launch that computes nothing."""

    def _sync_counter(self):
        pending = getattr(self, "_nbt_pending", 0)
        if pending and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(pending)
        self._nbt_pending = 0

    def forward(self, x):
        use_batch_stats = self.training or not self.track_running_stats
        if use_batch_stats and x.dim() == 2 and ops.bn_supported(x) and self.momentum is not None:
            rm = rv = None
            if self.training and self.track_running_stats:
                rm, rv = self.running_mean, self.running_var
                self._nbt_pending = getattr(self, "_nbt_pending", 0) + 1
            return ops.batch_norm_train(x, self.weight, self.bias, rm, rv, self.eps, self.momentum)
        self._sync_counter()
        return super().forward(x)

    def after_cgconv(self, conv, x, edge_index, edge_attr, csr, shift=None, packed=None):
        """self(conv(x, ...)) with the statistics in the conv kernel's epilogue (ops.cgconv_bn) when this module normalises with
        batch statistics and the layer has the shape for it; None otherwise (the caller composes)."""
        use_batch_stats = self.training or not self.track_running_stats
        if not (use_batch_stats and self.momentum is not None and self.num_features == x.shape[1] and ops.bn_supported(x)
                and ops.cgconv_bn_stats_ok(x, edge_attr, csr)):
            return None
        rm = rv = None
        if self.training and self.track_running_stats:
            rm, rv = self.running_mean, self.running_var
            self._nbt_pending = getattr(self, "_nbt_pending", 0) + 1
        return ops.cgconv_bn(x, edge_index, edge_attr, conv.lin_f.weight, conv.lin_f.bias, conv.lin_s.weight, conv.lin_s.bias,
                             conv.aggr, csr, self.weight, self.bias, rm, rv, self.eps, self.momentum, shift, packed)

    def after_linear_relu(self, h, weight, bias, lowp=None, gathered=None):
        """self(relu(F.linear(h, weight, bias) + gathered rows)) as one fused autograd node (ops.linear_relu_bn) when this
        module normalises with batch statistics and the layer has a fusable shape; None otherwise (the caller composes)."""
        use_batch_stats = self.training or not self.track_running_stats
        if not (use_batch_stats and self.momentum is not None and ops.linear_relu_bn_ok(h, weight, bias is not None, gathered)):
            return None
        rm = rv = None
        if self.training and self.track_running_stats:
            rm, rv = self.running_mean, self.running_var
            self._nbt_pending = getattr(self, "_nbt_pending", 0) + 1
        return ops.linear_relu_bn(h, weight, bias, lowp, self.weight, self.bias, rm, rv, self.eps, self.momentum, gathered)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._sync_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)
