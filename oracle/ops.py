"""Oracle operators: plain-PyTorch CPU restatement of the hot-path arithmetic (TEST INFRASTRUCTURE).

Every function cites the reference call site (paths relative to /root/reference) and, where the
arithmetic lives in an un-vendored third party (torch_geometric 2.0.1 / torch_scatter), the
published semantics it restates.  Only `index_select`, `index_add_`,
dense matmuls and elementwise ops are used, so the code runs on any torch build.

Convention (PyG flow="source_to_target"): edge_index[0] = source j ("row"),
edge_index[1] = target i ("col"); messages aggregate at the target.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------------
# scatter family — torch_scatter.scatter / scatter_mean as called at
# matdeeplearn/models/megnet.py:86,130-132,342-348 and inside every PyG propagate()/pool.
# --------------------------------------------------------------------------------------------
def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """out[index[k]] (+|max)= src[k] along dim 0.  mean divides by count.clamp(min=1); rows that
    receive nothing are 0 (also for max).  dim_size defaults to index.max()+1 (megnet.py:86)."""
    assert dim == 0, "the hot path only scatters along dim 0"
    n = (int(index.max()) + 1 if index.numel() else 0) if dim_size is None else int(dim_size)
    tail = tuple(src.shape[1:])
    if reduce in ("sum", "add", "mean"):
        out = torch.zeros((n,) + tail, dtype=src.dtype).index_add_(0, index, src)
        if reduce == "mean":
            cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
            out = out / cnt.clamp(min=1).view((n,) + (1,) * len(tail))
        return out
    if reduce == "max":
        # torch_scatter.scatter_max semantics: the value AND the gradient belong to ONE arg-max
        # element per (segment, channel) — the first one in index order on ties (CPU kernel: strict >).
        flat = src.reshape(src.shape[0], -1)
        idx = index.view(-1, 1).expand_as(flat)
        best = torch.full((n, flat.shape[1]), float("-inf"), dtype=src.dtype)
        best = best.scatter_reduce(0, idx, flat.detach(), "amax", include_self=True)
        pos = torch.arange(flat.shape[0]).view(-1, 1).expand_as(flat)
        cand = torch.where(flat.detach() == best.index_select(0, index), pos, torch.full_like(pos, flat.shape[0]))
        arg = torch.full((n, flat.shape[1]), flat.shape[0], dtype=torch.int64)
        arg = arg.scatter_reduce(0, idx, cand, "amin", include_self=True)
        has = arg < flat.shape[0]
        picked = flat.gather(0, arg.clamp(max=max(flat.shape[0] - 1, 0))) if flat.shape[0] else best
        out = torch.where(has, picked, torch.zeros_like(picked))
        return out.reshape((n,) + tail)
    raise ValueError("unsupported reduce: %r" % (reduce,))


def scatter_mean(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim, dim_size, "mean")


def scatter_add(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim, dim_size, "sum")


def global_mean_pool(x, batch, size=None):
    """torch_geometric.nn.global_mean_pool as looked up at matdeeplearn/models/cgcnn.py:154."""
    return scatter(x, batch, 0, size, "mean")


def global_add_pool(x, batch, size=None):
    return scatter(x, batch, 0, size, "sum")


def global_max_pool(x, batch, size=None):
    return scatter(x, batch, 0, size, "max")


POOLS = {"global_mean_pool": global_mean_pool, "global_add_pool": global_add_pool,
         "global_max_pool": global_max_pool}


# --------------------------------------------------------------------------------------------
# Gaussian RBF edge expansion — matdeeplearn/process/process.py:580-590 (class),
# :497-513 (call site, GaussianSmearing(0, 1, graph_edge_length, 0.2)), :626-653 (normalisation)
# --------------------------------------------------------------------------------------------
def rbf_offsets(start=0.0, stop=1.0, resolution=50):
    """process.py:583 — the centre grid is a float32 torch.linspace buffer."""
    return torch.linspace(start, stop, resolution)


def rbf_coeff(start=0.0, stop=1.0, width=0.2):
    """process.py:585 — coeff = -0.5 / ((stop - start) * width)**2, a Python float (−12.5)."""
    return -0.5 / ((stop - start) * width) ** 2


def rbf_expand(dist, start=0.0, stop=1.0, resolution=50, width=0.2):
    """process.py:588-590 — exp(coeff * (d[:,None] - offset[None,:])**2), fp32 in, fp32 out."""
    offset = rbf_offsets(start, stop, resolution).to(dist.dtype)
    diff = dist.unsqueeze(-1) - offset.view(1, -1)
    return torch.exp(rbf_coeff(start, stop, width) * torch.pow(diff, 2))


def edge_ranges(dist_list):
    """process.py:626-643 (GetRanges) — dataset-global (min, max) over per-graph distance tensors."""
    fmin = fmax = None
    for d in dist_list:
        if len(d) > 0:
            lo, hi = d.min(), d.max()
            fmin = lo if fmin is None or lo < fmin else fmin
            fmax = hi if fmax is None or hi > fmax else fmax
    return fmin, fmax


def normalize_edges(dist_list):
    """process.py:647-653 (NormalizeEdge) — (d - min) / (max - min) with the global range."""
    fmin, fmax = edge_ranges(dist_list)
    return [(d - fmin) / (fmax - fmin) for d in dist_list]


# --------------------------------------------------------------------------------------------
# CGConv — torch_geometric.nn.CGConv (2.0.1) as constructed at matdeeplearn/models/cgcnn.py:80-83
# (aggr="mean", batch_norm=False) and called at cgcnn.py:136-145.   [parity unpinned]
# --------------------------------------------------------------------------------------------
def cgconv(x, edge_index, edge_attr, w_f, b_f, w_s, b_s, aggr="mean"):
    """z = [x_i | x_j | e]; m = sigmoid(lin_f z) * softplus(lin_s z); out = aggr_{j->i} m + x."""
    row, col = edge_index[0], edge_index[1]
    z = torch.cat([x.index_select(0, col), x.index_select(0, row), edge_attr], dim=1)
    m = torch.sigmoid(F.linear(z, w_f, b_f)) * F.softplus(F.linear(z, w_s, b_s))
    return scatter(m, col, 0, x.shape[0], aggr) + x


class CGConv(nn.Module):
    """Parameter names/shapes per SURVEY A.2/A.7: lin_f, lin_s = Linear(2C+G, C)."""

    def __init__(self, channels, dim=0, aggr="mean", batch_norm=False, bias=True):
        super().__init__()
        assert not batch_norm, "the reference always passes batch_norm=False (cgcnn.py:81)"
        self.channels, self.dim, self.aggr = channels, dim, aggr
        self.lin_f = nn.Linear(2 * channels + dim, channels, bias=bias)
        self.lin_s = nn.Linear(2 * channels + dim, channels, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        """Upstream CGConv.__init__ ends with reset_parameters(): both Linears are drawn a second time (this matters
        only for reproducing a SEEDED initialisation; restated from the published 2.0.1 source, unverifiable here)."""
        self.lin_f.reset_parameters()
        self.lin_s.reset_parameters()

    def forward(self, x, edge_index, edge_attr):
        return cgconv(x, edge_index, edge_attr, self.lin_f.weight, self.lin_f.bias,
                      self.lin_s.weight, self.lin_s.bias, self.aggr)


# --------------------------------------------------------------------------------------------
# SchNet InteractionBlock / CFConv / ShiftedSoftplus — torch_geometric.nn.models.schnet (2.0.1)
# as constructed at matdeeplearn/models/schnet.py:81 and called at schnet.py:134-143. [unpinned]
# --------------------------------------------------------------------------------------------
class ShiftedSoftplus(nn.Module):
    def __init__(self):
        super().__init__()
        self.shift = math.log(2.0)

    def forward(self, x):
        return F.softplus(x) - self.shift


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

    def forward(self, x, edge_index, edge_weight, edge_attr):
        row, col = edge_index[0], edge_index[1]
        c = 0.5 * (torch.cos(edge_weight * math.pi / self.cutoff) + 1.0)
        w = self.nn(edge_attr) * c.view(-1, 1)
        h = self.lin1(x)
        agg = scatter(h.index_select(0, row) * w, col, 0, x.shape[0], "sum")
        return self.lin2(agg)


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
        """Upstream order: mlp[0], mlp[2], conv (lin1, lin2 — drawn a second time), lin."""
        for m in (self.mlp[0], self.mlp[2]):
            nn.init.xavier_uniform_(m.weight)
            m.bias.data.fill_(0)
        self.conv.reset_parameters()
        nn.init.xavier_uniform_(self.lin.weight)
        self.lin.bias.data.fill_(0)

    def forward(self, x, edge_index, edge_weight, edge_attr):
        return self.lin(self.act(self.conv(x, edge_index, edge_weight, edge_attr)))


# --------------------------------------------------------------------------------------------
# NNConv — torch_geometric.nn.NNConv (2.0.1) at matdeeplearn/models/mpnn.py:83-88,148-157. [unpinned]
# --------------------------------------------------------------------------------------------
class NNConv(nn.Module):
    def __init__(self, in_channels, out_channels, net, aggr="mean", root_weight=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.nn = net
        self.lin = nn.Linear(in_channels, out_channels, bias=False) if root_weight else None
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        """Upstream NNConv.__init__ ends with reset_parameters(): reset(self.nn) re-draws every layer of the edge
        network, then the root weight (PyG Linear 'uniform' = U(+-1/sqrt(fan_in)) = torch's default), bias = 0."""
        for m in self.nn.modules():
            if m is not self.nn and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        if self.lin is not None:
            self.lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, edge_index, edge_attr):
        row, col = edge_index[0], edge_index[1]
        w = self.nn(edge_attr).view(-1, self.in_channels, self.out_channels)
        m = torch.matmul(x.index_select(0, row).unsqueeze(1), w).squeeze(1)
        out = scatter(m, col, 0, x.shape[0], self.aggr)
        if self.lin is not None:
            out = out + self.lin(x)
        if self.bias is not None:
            out = out + self.bias
        return out


# --------------------------------------------------------------------------------------------
# GCNConv — torch_geometric.nn.GCNConv (2.0.1) at matdeeplearn/models/gcn.py:80-82,135-144
# (improved=True, add_self_loops=False, edge_weight = raw distance). [unpinned]
# --------------------------------------------------------------------------------------------
class GCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, add_self_loops=True, bias=True):
        super().__init__()
        assert not add_self_loops, "the reference passes add_self_loops=False (gcn.py:81)"
        # PyG's own Linear(weight_initializer="glorot") allocates its weight WITHOUT a draw and then draws glorot once in its
        # ctor; torch's nn.Linear would consume one more (kaiming) draw and shift every later layer's seeded initialisation
        self.lin = torch.nn.utils.skip_init(nn.Linear, in_channels, out_channels, bias=False)
        nn.init.xavier_uniform_(self.lin.weight)            # ... the ctor's draw ...
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()                             # ... and GCNConv.__init__ ends with reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.lin.weight)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, edge_index, edge_weight=None):
        row, col = edge_index[0], edge_index[1]
        n = x.shape[0]
        if edge_weight is None:
            edge_weight = torch.ones(row.numel(), dtype=x.dtype)
        deg = scatter(edge_weight, col, 0, n, "sum")
        dis = deg.pow(-0.5)
        dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
        norm = dis.index_select(0, row) * edge_weight * dis.index_select(0, col)
        h = self.lin(x)
        out = scatter(norm.view(-1, 1) * h.index_select(0, row), col, 0, n, "sum")
        return out + self.bias if self.bias is not None else out


# --------------------------------------------------------------------------------------------
# MetaLayer — torch_geometric.nn.MetaLayer (2.0.1) at matdeeplearn/models/megnet.py:235-253.
# Pinned through the MEGNet goldens (the golden generator uses the same published semantics).
# --------------------------------------------------------------------------------------------
class MetaLayer(nn.Module):
    def __init__(self, edge_model=None, node_model=None, global_model=None):
        super().__init__()
        self.edge_model, self.node_model, self.global_model = edge_model, node_model, global_model

    def forward(self, x, edge_index, edge_attr=None, u=None, batch=None):
        row, col = edge_index[0], edge_index[1]
        if self.edge_model is not None:
            edge_attr = self.edge_model(x.index_select(0, row), x.index_select(0, col), edge_attr, u,
                                        batch if batch is None else batch.index_select(0, row))
        if self.node_model is not None:
            x = self.node_model(x, edge_index, edge_attr, u, batch)
        if self.global_model is not None:
            u = self.global_model(x, edge_index, edge_attr, u, batch)
        return x, edge_attr, u


# --------------------------------------------------------------------------------------------
# Set2Set — torch_geometric.nn.Set2Set (2.0.1) at matdeeplearn/models/cgcnn.py:112-119,152. [unpinned]
# --------------------------------------------------------------------------------------------
class Set2Set(nn.Module):
    def __init__(self, in_channels, processing_steps, num_layers=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = nn.LSTM(self.out_channels, in_channels, num_layers)
        self.lstm.reset_parameters()                        # upstream Set2Set.__init__ ends with reset_parameters()

    def forward(self, x, batch):
        b = int(batch.max()) + 1
        h = (x.new_zeros((self.num_layers, b, self.in_channels)),
             x.new_zeros((self.num_layers, b, self.in_channels)))
        q_star = x.new_zeros(b, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self.lstm(q_star.unsqueeze(0), h)
            q = q.view(b, self.in_channels)
            e = (x * q.index_select(0, batch)).sum(dim=-1, keepdim=True)
            emax = scatter(e, batch, 0, b, "max").index_select(0, batch)
            a = torch.exp(e - emax)
            a = a / (scatter(a, batch, 0, b, "sum").index_select(0, batch) + 1e-16)
            r = scatter(a * x, batch, 0, b, "sum")
            q_star = torch.cat([q, r], dim=-1)
        return q_star


# --------------------------------------------------------------------------------------------
# Independent dense-adjacency derivations (fp64) used by tests/test_oracle_selfcheck.py
# --------------------------------------------------------------------------------------------
def cgconv_dense(x, edge_index, edge_attr, w_f, b_f, w_s, b_s):
    """Per-target python loop, no scatter: an independent restatement of A.2 for cross-checking."""
    n = x.shape[0]
    out = x.clone()
    row, col = edge_index[0].tolist(), edge_index[1].tolist()
    for i in range(n):
        msgs = []
        for e, (j, t) in enumerate(zip(row, col)):
            if t != i:
                continue
            z = torch.cat([x[i], x[j], edge_attr[e]])
            f = w_f @ z + b_f
            s = w_s @ z + b_s
            msgs.append((1.0 / (1.0 + torch.exp(-f))) * torch.log1p(torch.exp(s)))
        if msgs:
            out[i] = out[i] + torch.stack(msgs).mean(0)
    return out


def cfconv_loop(x, edge_index, edge_weight, edge_attr, mlp0_w, mlp0_b, mlp2_w, mlp2_b, lin1_w, lin2_w, lin2_b, lin_w, lin_b, cutoff):
    """InteractionBlock (A.3) edge by edge, no scatter, no nn.Module: filter W_e = mlp2(ssp(mlp0 e_e)) * C(d_e) with the cosine
    cutoff C(d) = (cos(d pi / cutoff) + 1) / 2, message h_j * W_e with h = lin1 x (no bias), sum at the TARGET, then
    lin(ssp(lin2 agg)).  An independent restatement for tests/test_oracle_selfcheck.py."""
    ssp = lambda t: torch.log1p(torch.exp(t)) - math.log(2.0)
    n = x.shape[0]
    h = x @ lin1_w.T
    agg = torch.zeros(n, lin1_w.shape[0], dtype=x.dtype)
    for e in range(edge_index.shape[1]):
        j, i = int(edge_index[0, e]), int(edge_index[1, e])
        w = mlp2_w @ ssp(mlp0_w @ edge_attr[e] + mlp0_b) + mlp2_b
        agg[i] = agg[i] + h[j] * w * (0.5 * (math.cos(float(edge_weight[e]) * math.pi / cutoff) + 1.0))
    return ssp(agg @ lin2_w.T + lin2_b) @ lin_w.T + lin_b


def nnconv_loop(x, edge_index, edge_attr, nn0_w, nn0_b, nn2_w, nn2_b, root_w, bias, aggr="mean"):
    """NNConv (A.4) edge by edge: Theta_e = reshape(nn2 relu(nn0 e_e), [C_in, C_out]) (row-major .view), message x_j^T Theta_e,
    mean over the incoming edges of the TARGET (0 for none), + root_w x_i + bias."""
    n, c_in = x.shape
    c_out = root_w.shape[0]
    out = x @ root_w.T + bias
    for i in range(n):
        msgs = []
        for e in range(edge_index.shape[1]):
            if int(edge_index[1, e]) != i:
                continue
            theta = (nn2_w @ torch.clamp(nn0_w @ edge_attr[e] + nn0_b, min=0) + nn2_b).reshape(c_in, c_out)
            msgs.append(x[int(edge_index[0, e])] @ theta)
        if msgs:
            s = torch.stack(msgs).sum(0)
            out[i] = out[i] + (s / len(msgs) if aggr == "mean" else s)
    return out


def gcnconv_dense(x, edge_index, edge_weight, lin_w, bias):
    """GCNConv(add_self_loops=False) (A.5) as dense matrices: A[i, j] = sum of the weights of the edges j -> i, weighted
    in-degree deg_i = sum_j A[i, j], out = D^-1/2 A D^-1/2 (x W^T) + bias with 0 for deg = 0 (inf -> 0)."""
    n = x.shape[0]
    A = torch.zeros(n, n, dtype=x.dtype)
    for e in range(edge_index.shape[1]):
        A[int(edge_index[1, e]), int(edge_index[0, e])] += edge_weight[e]
    deg = A.sum(1)
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    return (dis[:, None] * A * dis[None, :]) @ (x @ lin_w.T) + bias


def set2set_loop(x, batch, w_ih, w_hh, b_ih, b_hh, steps):
    """Set2Set (A.6) graph by graph with the one-layer LSTM cell written out (gate order i, f, g, o of torch.nn.LSTM):
    q* = 0; repeat: (q, c) = LSTMCell(q*, (q, c)); a_n = softmax over the graph's nodes of <x_n, q>; r = sum a_n x_n;
    q* = [q | r].  No scatter, no nn.LSTM."""
    b = int(batch.max()) + 1
    C = x.shape[1]
    out = []
    for g in range(b):
        xs = x[batch == g]
        hq = torch.zeros(C, dtype=x.dtype)
        c = torch.zeros(C, dtype=x.dtype)
        q_star = torch.zeros(2 * C, dtype=x.dtype)
        for _ in range(steps):
            gates = w_ih @ q_star + b_ih + w_hh @ hq + b_hh
            i_, f_, g_, o_ = gates[:C], gates[C:2 * C], gates[2 * C:3 * C], gates[3 * C:]
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            hq = torch.sigmoid(o_) * torch.tanh(c)
            e = xs @ hq
            a = torch.exp(e - e.max())
            a = a / a.sum()
            q_star = torch.cat([hq, (a[:, None] * xs).sum(0)])
        out.append(q_star)
    return torch.stack(out)
