"""MEGNet — /root/reference/matdeeplearn/models/megnet.py:16-371.  Edge / node / global blocks
(:16-56, :59-101, :104-147): Linear -> act -> BatchNorm1d -> dropout (act BEFORE BN), scatter_mean of
the edge state at the SOURCE row (:86,130), MetaLayer wiring (:235-253), residual rule (first layer adds
the embedded inputs, later layers the running state, :313-336), 3-way pooling [x | e | u] (:339-349).
Gathers and scatters run on the HIP kernels.  In bf16 mode every Linear is the streaming HIP dense layer (forward) +
TN GEMM (dW), and the edge block's first layer is K6: the [E, 4d] concatenation [x[row] | x[col] | e | u[batch]] of the
reference is never built — the weight is split by column blocks into three per-node / per-graph projections and one K = d
product over the edge state whose epilogue adds the gathered projection rows (csrc/linear.hip).  fp32 (parity) mode keeps
the reference's formulation on library GEMMs."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..nn import BatchNorm1d, MetaLayer, Set2Set, _seq
from ._base import GraphModel, _lowp, dense, dense_act


class _Mlp(nn.Module):
    def __init__(self, list_name, in_dim, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers):
        super().__init__()
        self.act, self.batch_norm, self.dropout_rate, self.list_name = act, batch_norm, dropout_rate, list_name
        track = batch_track_stats != "False"
        setattr(self, list_name, nn.ModuleList(
            [nn.Linear(in_dim if i == 0 else dim, dim) for i in range(fc_layers + 1)]))
        self.bn_list = nn.ModuleList(
            [BatchNorm1d(dim, track_running_stats=track) for _ in range(fc_layers + 1)] if batch_norm == "True" else [])

    def _tail(self, out, first, normed=False):
        """BatchNorm / dropout of layer `first` (normed: its BatchNorm has been applied already), then the remaining
        Linear -> act -> BatchNorm -> dropout layers; in bf16 a Linear -> ReLU -> BatchNorm triple with batch statistics is one
        fused autograd node (BatchNorm1d.after_linear_relu)."""
        lins = getattr(self, self.list_name)
        for i in range(first, len(lins)):
            done = normed and i == first
            if i > first:
                z = None
                if self.batch_norm == "True" and self.act == "relu" and out.dtype != lins[i].weight.dtype:
                    z = self.bn_list[i].after_linear_relu(out, lins[i].weight, lins[i].bias, _lowp(lins[i]))
                if z is not None:
                    out, done = z, True
                else:
                    out = dense_act(lins[i], out, self.act)
            if self.batch_norm == "True" and not done:
                out = self.bn_list[i](out)
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        return out

    def run(self, comb):
        return self._tail(dense_act(getattr(self, self.list_name)[0], comb, self.act), 0)


class Megnet_EdgeModel(_Mlp):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("edge_mlp", dim * 4, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, src, dest, edge_attr, u, batch):
        return self.run(torch.cat([src, dest, edge_attr, ops.gather(u, batch)], dim=1))

    def fused_ok(self, x, edge_attr):
        d = self.edge_mlp[0].out_features
        return (edge_attr.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and self.act == "relu" and d <= 128
                and d % 2 == 0 and edge_attr.is_contiguous() and edge_attr.data_ptr() % 16 == 0)

    def forward_fused(self, x, row, col, edge_attr, u, batch_n):
        """K6: same function as forward(x[row], x[col], edge_attr, u, batch_e) without the gathers and the concatenation."""
        lin = self.edge_mlp[0]
        d = lin.out_features
        cd = x.dtype
        wa, wb, wc, wd = (lin.weight[:, k * d:(k + 1) * d] for k in range(4))       # column blocks: src | dest | e | u
        p_src = ops.linear(x, wa, None)                                             # [N, d] per-node projections (dW: TN GEMM)
        p_glb = F.linear(u, wd.to(cd), None if lin.bias is None else lin.bias.to(cd))  # [B, d] per graph (+ bias)
        # batch[row] == batch[col] (an edge stays inside its graph): the per-graph row is folded into the TARGET node's row, so
        # the kernel gathers two tables, and the per-graph gradient is a 25-row reduction over nodes instead of a
        # 325-row one over edges
        p_dst = ops.linear(x, wb, None) + ops.gather(p_glb, batch_n)
        if self.batch_norm == "True":
            z = self.bn_list[0].after_linear_relu(edge_attr, wc, None, None, [(p_src, row), (p_dst, col)])
            if z is not None:
                return self._tail(z, 0, normed=True)
        out = ops.linear_gather_act(edge_attr, wc, None, self.act, [(p_src, row), (p_dst, col)])
        return self._tail(out, 0)


class Megnet_NodeModel(_Mlp):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("node_mlp", dim * 3, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, x, edge_index, edge_attr, u, batch, v_e=None):
        if v_e is None:
            v_e = ops.scatter_mean(edge_attr, edge_index[0], 0, x.shape[0])      # aggregate at the SOURCE row
        return self.run(torch.cat([x, v_e, ops.gather(u, batch)], dim=1))


class Megnet_GlobalModel(_Mlp):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("global_mlp", dim * 3, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, x, edge_index, edge_attr, u, batch, v_e=None):
        b = u.shape[0]
        if v_e is None:
            v_e = ops.scatter_mean(edge_attr, edge_index[0], 0, x.shape[0])
        u_e = ops.scatter_mean(v_e, batch, 0, b, assume_sorted=True)
        u_v = ops.scatter_mean(x, batch, 0, b, assume_sorted=True)
        return self.run(torch.cat([u_e, u_v, u], dim=1))


def _embed(i, d):
    return nn.Sequential(nn.Linear(i, d), nn.ReLU(), nn.Linear(d, d), nn.ReLU())


def _run_embed(seq, h):
    """Sequential(Linear, ReLU, Linear, ReLU) (megnet.py:222-247) with each Linear + ReLU pair as one fused dense layer and
    the first ReLU's derivative handed to the second layer's backward (nn._seq)."""
    return _seq(seq, h)


class _FusedMetaLayer(MetaLayer):
    """MetaLayer (megnet.py:235-253) that hands the edge model the UNGATHERED node state when it can fuse the gathers."""

    def forward(self, x, edge_index, edge_attr=None, u=None, batch=None, idx=None, e_res=None):
        """e_res: the edge state the caller adds to this layer's edge output (megnet.py:321-336); when given, the returned
        edge tensor is that SUM (formed with the by-source mean of the output in one autograd node)."""
        em = self.edge_model
        if idx is not None and em is not None and hasattr(em, "forward_fused"):
            row32, col32, batch_n = idx
            if em.fused_ok(x, edge_attr):
                edge_attr = em.forward_fused(x, row32, col32, edge_attr, u, batch_n)
            else:
                # the reference's formulation with the loader's index tensors: u[batch[row]] as two gathers, so that every
                # gather's backward is a scatter over an index whose segments the loader already knows (no sort; and in a
                # padded static batch no segment made of the unused edge slots)
                edge_attr = em.run(torch.cat([ops.gather(x, row32), ops.gather(x, col32), edge_attr,
                                              ops.gather(ops.gather(u, batch_n), row32)], dim=1))
            # the node block and the global block both start from scatter_mean(e', edge_index[0]) (megnet.py:86 and :130): the
            # same [E, d] -> [N, d] reduction of the same tensor, formed ONCE here (one segmented reduction forward, one backward
            # and one gradient accumulation over the edge rows less per block)
            v_e, e_out = None, None
            if self.node_model is not None and self.global_model is not None:
                if e_res is not None:
                    e_out, v_e = ops.residual_scatter(edge_attr, e_res, edge_index[0], x.shape[0], "mean")
                else:
                    v_e = ops.scatter_mean(edge_attr, edge_index[0], 0, x.shape[0])
            if self.node_model is not None:
                x = self.node_model(x, edge_index, edge_attr, u, batch, v_e=v_e)
            if self.global_model is not None:
                u = self.global_model(x, edge_index, edge_attr, u, batch, v_e=v_e)
            if e_res is not None and e_out is None:
                e_out = edge_attr + e_res
            return x, (edge_attr if e_res is None else e_out), u
        x, e, u = super().forward(x, edge_index, edge_attr, u, batch)
        return x, (e if e_res is None else e + e_res), u


class MEGNet(GraphModel):
    def __init__(self, data, dim1=64, dim2=64, dim3=64, pre_fc_count=1, gc_count=3, gc_fc_count=2,
                 post_fc_count=1, pool="global_mean_pool", pool_order="early", batch_norm="True",
                 batch_track_stats="True", act="relu", dropout_rate=0.0, compute_dtype="fp32", **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, compute_dtype,
                    lists=("e_embed_list", "x_embed_list", "u_embed_list", "conv_list", "bn_list"))
        self.pool_reduce = {"global_mean_pool": "mean", "global_max_pool": "max", "global_sum_pool": "sum"}.get(pool)
        for i in range(gc_count):
            self.e_embed_list.append(_embed(data.num_edge_features if i == 0 else dim3, dim3))
            self.x_embed_list.append(_embed(self.gc_dim if i == 0 else dim3, dim3))
            self.u_embed_list.append(_embed(data[0].u.shape[1] if i == 0 else dim3, dim3))
            args = (dim3, act, batch_norm, batch_track_stats, dropout_rate, gc_fc_count)
            self.conv_list.append(_FusedMetaLayer(Megnet_EdgeModel(*args), Megnet_NodeModel(*args), Megnet_GlobalModel(*args)))
        self._finish(dim2, post_fc_count, dim3, early_mult=3, set2set_names=("set2set_x", "set2set_e"))

    def forward(self, data):
        cd = self.compute_dtype
        out = self._pre(data.x.to(cd))
        ei = data.edge_index
        nb = getattr(data, "num_graphs", None) or data.u.shape[0]
        idx = None
        csr = getattr(data, "csr", None)
        if csr is not None and csr.eperm is None:    # product loader: the int32 source / target arrays ARE the gather indices
            idx = (csr.row, csr.col, data.batch)
        elif cd == torch.bfloat16:                   # int32 gather indices of the fused edge block, once per batch
            idx = (ei[0].to(torch.int32), ei[1].to(torch.int32), data.batch)
        x = e = u = None
        for i, conv in enumerate(self.conv_list):
            e_t = _run_embed(self.e_embed_list[i], data.edge_attr.to(cd) if i == 0 else e)
            x_t = _run_embed(self.x_embed_list[i], out if i == 0 else x)
            u_t = _run_embed(self.u_embed_list[i], data.u.to(cd) if i == 0 else u)
            # (the edge residual is formed inside the layer, together with the by-source mean of its edge output)
            x_o, e, u_o = conv(x_t, ei, e_t, u_t, data.batch, idx=idx, e_res=e_t if i == 0 else e)
            if i == 0:
                x, u = x_o + x_t, u_o + u_t
            else:
                x, u = x_o + x, u_o + u
        n = x.shape[0]
        if self.pool_order == "early":
            if self.pool == "set2set":
                x_pool = self.set2set_x(x, data.batch, nb)
                e_pool = self.set2set_e(ops.scatter(e, ei[0], 0, n, "mean"), data.batch, nb)
            else:
                x_pool = ops.scatter(x, data.batch, 0, nb, self.pool_reduce, assume_sorted=True)
                e_pool = ops.scatter(ops.scatter(e, ei[0], 0, n, self.pool_reduce), data.batch, 0, nb, self.pool_reduce,
                                     assume_sorted=True)
            out = self._post(torch.cat([x_pool, e_pool, u], dim=1), final=True)
        else:
            out = self._post(x)
            if self.pool == "set2set":
                out = dense(self.lin_out_2, self.set2set_x(out, data.batch, nb))
            else:
                out = ops.POOLS[self.pool](out, data.batch, nb)
        out = out.float()
        return out.view(-1) if out.shape[1] == 1 else out
