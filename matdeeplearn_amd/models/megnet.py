"""MEGNet — /root/reference/matdeeplearn/models/megnet.py:16-371.  Edge / node / global blocks
(:16-56, :59-101, :104-147): Linear -> act -> BatchNorm1d -> dropout (act BEFORE BN), scatter_mean of
the edge state at the SOURCE row (:86,130), MetaLayer wiring (:235-253), residual rule (first layer adds
the embedded inputs, later layers the running state, :313-336), 3-way pooling [x | e | u] (:339-349).
Gathers and scatters run on the HIP kernels; the MLPs are library GEMMs."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..nn import BatchNorm1d, MetaLayer, Set2Set
from ._base import GraphModel, dense


class _Mlp(nn.Module):
    def __init__(self, list_name, in_dim, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers):
        super().__init__()
        self.act, self.batch_norm, self.dropout_rate, self.list_name = act, batch_norm, dropout_rate, list_name
        track = batch_track_stats != "False"
        setattr(self, list_name, nn.ModuleList(
            [nn.Linear(in_dim if i == 0 else dim, dim) for i in range(fc_layers + 1)]))
        self.bn_list = nn.ModuleList(
            [BatchNorm1d(dim, track_running_stats=track) for _ in range(fc_layers + 1)] if batch_norm == "True" else [])

    def run(self, comb):
        out = comb
        for i, lin in enumerate(getattr(self, self.list_name)):
            out = getattr(F, self.act)(dense(lin, out))
            if self.batch_norm == "True":
                out = self.bn_list[i](out)
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        return out


class Megnet_EdgeModel(_Mlp):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("edge_mlp", dim * 4, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, src, dest, edge_attr, u, batch):
        return self.run(torch.cat([src, dest, edge_attr, ops.gather(u, batch)], dim=1))


class Megnet_NodeModel(_Mlp):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("node_mlp", dim * 3, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, x, edge_index, edge_attr, u, batch):
        v_e = ops.scatter_mean(edge_attr, edge_index[0], 0, x.shape[0])      # aggregate at the SOURCE row
        return self.run(torch.cat([x, v_e, ops.gather(u, batch)], dim=1))


class Megnet_GlobalModel(_Mlp):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("global_mlp", dim * 3, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, x, edge_index, edge_attr, u, batch):
        b = u.shape[0]
        u_e = ops.scatter_mean(ops.scatter_mean(edge_attr, edge_index[0], 0, x.shape[0]), batch, 0, b, assume_sorted=True)
        u_v = ops.scatter_mean(x, batch, 0, b, assume_sorted=True)
        return self.run(torch.cat([u_e, u_v, u], dim=1))


def _embed(i, d):
    return nn.Sequential(nn.Linear(i, d), nn.ReLU(), nn.Linear(d, d), nn.ReLU())


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
            self.conv_list.append(MetaLayer(Megnet_EdgeModel(*args), Megnet_NodeModel(*args), Megnet_GlobalModel(*args)))
        self._finish(dim2, post_fc_count, dim3, early_mult=3, set2set_names=("set2set_x", "set2set_e"))

    def forward(self, data):
        cd = self.compute_dtype
        out = self._pre(data.x.to(cd))
        ei = data.edge_index
        nb = getattr(data, "num_graphs", None) or data.u.shape[0]
        x = e = u = None
        for i, conv in enumerate(self.conv_list):
            e_t = self.e_embed_list[i](data.edge_attr.float() if i == 0 else e)
            x_t = self.x_embed_list[i](out.float() if i == 0 else x)
            u_t = self.u_embed_list[i](data.u.float() if i == 0 else u)
            x_o, e_o, u_o = conv(x_t, ei, e_t, u_t, data.batch)
            if i == 0:
                x, e, u = x_o + x_t, e_o + e_t, u_o + u_t
            else:
                x, e, u = x_o + x, e_o + e, u_o + u
        n = x.shape[0]
        if self.pool_order == "early":
            if self.pool == "set2set":
                x_pool = self.set2set_x(x, data.batch, nb)
                e_pool = self.set2set_e(ops.scatter(e, ei[0], 0, n, "mean"), data.batch, nb)
            else:
                x_pool = ops.scatter(x, data.batch, 0, nb, self.pool_reduce, assume_sorted=True)
                e_pool = ops.scatter(ops.scatter(e, ei[0], 0, n, self.pool_reduce), data.batch, 0, nb, self.pool_reduce,
                                     assume_sorted=True)
            out = self._post(torch.cat([x_pool, e_pool, u], dim=1))
        else:
            out = self._post(x)
            if self.pool == "set2set":
                out = dense(self.lin_out_2, self.set2set_x(out, data.batch, nb))
            else:
                out = ops.POOLS[self.pool](out, data.batch, nb)
        out = out.float()
        return out.view(-1) if out.shape[1] == 1 else out
