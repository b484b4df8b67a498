"""Oracle model wrappers (TEST INFRASTRUCTURE): CPU restatement of matdeeplearn/models/*.py.

One shared skeleton replaces the reference's five near-identical files:
  pre-FC (+act)  ->  gc_count x [conv -> BatchNorm1d -> (act) -> dropout]  ->  pool  ->  post-FC -> lin_out
with the per-architecture differences kept exactly as the reference has them:
  CGCNN  matdeeplearn/models/cgcnn.py:121-174   conv=CGConv(aggr=mean), NO activation between layers (:146)
  SchNet matdeeplearn/models/schnet.py:121-172  out = out + InteractionBlock(...), BN after the residual, no act
  GCN    matdeeplearn/models/gcn.py:120-173     GCNConv(edge_weight = raw distance) -> BN -> act
  MPNN   matdeeplearn/models/mpnn.py:129-188    NNConv -> BN -> act -> dropout -> one GRU step (h0 = pre-FC out)
  MEGNet matdeeplearn/models/megnet.py:150-371  embed MLPs -> MetaLayer(edge,node,global) -> residuals -> 3-way pool
State-dict key names follow SURVEY.md Appendix A.7 so weights interchange with the product models.
String booleans ("True"/"False") are kept (config.yml:129-131).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def _out_dim(data):
    y = data[0].y
    return 1 if y.ndim == 0 else len(y[0])


class _Skeleton(nn.Module):
    """Everything the five wrappers share (e.g. cgcnn.py:35-119).  Layers are CREATED in the reference's order —
    pre_lin_list -> per conv layer (conv [, gru], bn) -> post_lin_list -> lin_out -> set2set [-> lin_out_2]
    (cgcnn.py:64-119, mpnn.py:66-128) — and the ModuleLists are REGISTERED in the reference's order, so that a
    seeded construction draws the same initial weights and state_dict() lists the same keys in the same order
    (pinned by tests/golden/wrappers.npz, generated from the reference files themselves)."""

    def _begin(self, data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
               dropout_rate, lists=("conv_list", "bn_list")):
        assert gc_count > 0, "Need at least 1 GC layer"
        self.batch_track_stats = batch_track_stats != "False"
        self.batch_norm, self.pool, self.act = batch_norm, pool, act
        self.pool_order, self.dropout_rate = pool_order, dropout_rate
        self.gc_dim = data.num_features if pre_fc_count == 0 else dim1
        self.output_dim = _out_dim(data)
        self.pre_lin_list = nn.ModuleList(
            [nn.Linear(data.num_features if i == 0 else dim1, dim1) for i in range(pre_fc_count)])
        for name in lists:
            setattr(self, name, nn.ModuleList())

    def _add_bn(self, dim):
        if self.batch_norm == "True":
            self.bn_list.append(nn.BatchNorm1d(dim, track_running_stats=self.batch_track_stats))

    def _finish(self, dim2, post_fc_count, post_in, pool_mult=1, set2set_names=("set2set",)):
        early_s2s = self.pool_order == "early" and self.pool == "set2set"
        if self.pool_order == "early":
            first_in = post_in * (pool_mult * 2 - 1 if (early_s2s and pool_mult > 1) else
                                  (2 if early_s2s else pool_mult))
        else:
            first_in = post_in
        self.post_lin_list = nn.ModuleList(
            [nn.Linear(first_in if i == 0 else dim2, dim2) for i in range(post_fc_count)])
        self.lin_out = nn.Linear(dim2 if post_fc_count > 0 else first_in, self.output_dim)
        if early_s2s:
            for name in set2set_names:
                setattr(self, name, ops.Set2Set(post_in, processing_steps=3))
        elif self.pool == "set2set" and self.pool_order == "late":
            setattr(self, set2set_names[0], ops.Set2Set(self.output_dim, processing_steps=3, num_layers=1))
            self.lin_out_2 = nn.Linear(self.output_dim * 2, self.output_dim)

    def _pre(self, data):
        out = data.x
        for lin in self.pre_lin_list:
            out = getattr(F, self.act)(lin(out))
        return out

    def _bn(self, i, out):
        return self.bn_list[i](out) if self.batch_norm == "True" else out

    def _post(self, out):
        for lin in self.post_lin_list:
            out = getattr(F, self.act)(lin(out))
        return self.lin_out(out)

    def _head(self, out, batch):
        """cgcnn.py:149-174 — early / late pooling and the final view(-1)."""
        if self.pool_order == "early":
            out = self.set2set(out, batch) if self.pool == "set2set" else ops.POOLS[self.pool](out, batch)
            out = self._post(out)
        else:
            out = self._post(out)
            if self.pool == "set2set":
                out = self.lin_out_2(self.set2set(out, batch))
            else:
                out = ops.POOLS[self.pool](out, batch)
        return out.view(-1) if out.shape[1] == 1 else out


class CGCNN(_Skeleton):
    def __init__(self, data, dim1=64, dim2=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate)
        for _ in range(gc_count):                                                 # cgcnn.py:77-87
            self.conv_list.append(ops.CGConv(self.gc_dim, data.num_edge_features, aggr="mean", batch_norm=False))
            self._add_bn(self.gc_dim)
        self._finish(dim2, post_fc_count, self.gc_dim)

    def forward(self, data):
        out = self._pre(data)
        for i, conv in enumerate(self.conv_list):
            out = self._bn(i, conv(out, data.edge_index, data.edge_attr))
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        return self._head(out, data.batch)


class SchNet(_Skeleton):
    def __init__(self, data, dim1=64, dim2=64, dim3=64, cutoff=8, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate)
        for _ in range(gc_count):                                                 # schnet.py:78-86
            self.conv_list.append(ops.InteractionBlock(self.gc_dim, data.num_edge_features, dim3, cutoff))
            self._add_bn(self.gc_dim)
        self._finish(dim2, post_fc_count, self.gc_dim)

    def forward(self, data):
        out = self._pre(data)
        for i, conv in enumerate(self.conv_list):
            out = self._bn(i, out + conv(out, data.edge_index, data.edge_weight, data.edge_attr))
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        return self._head(out, data.batch)


class GCN(_Skeleton):
    def __init__(self, data, dim1=64, dim2=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate)
        for _ in range(gc_count):                                                 # gcn.py:77-86
            self.conv_list.append(ops.GCNConv(self.gc_dim, self.gc_dim, improved=True, add_self_loops=False))
            self._add_bn(self.gc_dim)
        self._finish(dim2, post_fc_count, self.gc_dim)

    def forward(self, data):
        out = self._pre(data)
        for i, conv in enumerate(self.conv_list):
            out = self._bn(i, conv(out, data.edge_index, data.edge_weight))
            out = getattr(F, self.act)(out)
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        return self._head(out, data.batch)


class MPNN(_Skeleton):
    def __init__(self, data, dim1=64, dim2=64, dim3=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, lists=("conv_list", "gru_list", "bn_list"))
        c = self.gc_dim
        for _ in range(gc_count):                                                 # mpnn.py:78-96
            net = nn.Sequential(nn.Linear(data.num_edge_features, dim3), nn.ReLU(), nn.Linear(dim3, c * c))
            self.conv_list.append(ops.NNConv(c, c, net, aggr="mean"))
            self.gru_list.append(nn.GRU(c, c))
            self._add_bn(c)
        self._finish(dim2, post_fc_count, c)

    def forward(self, data):
        out = self._pre(data)
        h = out.unsqueeze(0)
        for i, conv in enumerate(self.conv_list):
            m = self._bn(i, conv(out, data.edge_index, data.edge_attr))
            m = getattr(F, self.act)(m)
            m = F.dropout(m, p=self.dropout_rate, training=self.training)
            out, h = self.gru_list[i](m.unsqueeze(0), h)
            out = out.squeeze(0)
        return self._head(out, data.batch)


# ------------------------------------------------------------------------------------------------
# MEGNet — every block's arithmetic is in the reference tree; PINNED by tests/golden/megnet.npz.
# ------------------------------------------------------------------------------------------------
class _MegnetMLP(nn.Module):
    """Linear -> act -> BatchNorm1d -> dropout, (fc_layers + 1) times (megnet.py:28-56: act BEFORE BN)."""

    def __init__(self, list_name, in_dim, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers):
        super().__init__()
        self.act, self.batch_norm, self.dropout_rate, self.list_name = act, batch_norm, dropout_rate, list_name
        track = batch_track_stats != "False"
        setattr(self, list_name, nn.ModuleList(
            [nn.Linear(in_dim if i == 0 else dim, dim) for i in range(fc_layers + 1)]))
        self.bn_list = nn.ModuleList(
            [nn.BatchNorm1d(dim, track_running_stats=track) for _ in range(fc_layers + 1)]
            if batch_norm == "True" else [])

    def run(self, comb):
        out = comb
        for i, lin in enumerate(getattr(self, self.list_name)):
            out = getattr(F, self.act)(lin(out))
            if self.batch_norm == "True":
                out = self.bn_list[i](out)
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        return out


class MegnetEdgeModel(_MegnetMLP):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("edge_mlp", dim * 4, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, src, dest, edge_attr, u, batch):  # megnet.py:41-56
        return self.run(torch.cat([src, dest, edge_attr, u.index_select(0, batch)], dim=1))


class MegnetNodeModel(_MegnetMLP):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("node_mlp", dim * 3, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, x, edge_index, edge_attr, u, batch):  # megnet.py:84-101 (aggregate at SOURCE row)
        v_e = ops.scatter_mean(edge_attr, edge_index[0], 0)
        return self.run(torch.cat([x, v_e, u.index_select(0, batch)], dim=1))


class MegnetGlobalModel(_MegnetMLP):
    def __init__(self, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers=2):
        super().__init__("global_mlp", dim * 3, dim, act, batch_norm, batch_track_stats, dropout_rate, fc_layers)

    def forward(self, x, edge_index, edge_attr, u, batch):  # megnet.py:129-147
        u_e = ops.scatter_mean(ops.scatter_mean(edge_attr, edge_index[0], 0), batch, 0)
        u_v = ops.scatter_mean(x, batch, 0)
        return self.run(torch.cat([u_e, u_v, u], dim=1))


def _embed(i, d):
    return nn.Sequential(nn.Linear(i, d), nn.ReLU(), nn.Linear(d, d), nn.ReLU())


class MEGNet(_Skeleton):
    def __init__(self, data, dim1=64, dim2=64, dim3=64, pre_fc_count=1, gc_count=3, gc_fc_count=2,
                 post_fc_count=1, pool="global_mean_pool", pool_order="early", batch_norm="True",
                 batch_track_stats="True", act="relu", dropout_rate=0.0, **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, lists=("e_embed_list", "x_embed_list", "u_embed_list", "conv_list", "bn_list"))
        self.pool_reduce = {"global_mean_pool": "mean", "global_max_pool": "max",
                            "global_sum_pool": "sum"}.get(pool)  # megnet.py:177-182 (no global_add_pool)
        for i in range(gc_count):                                                 # megnet.py:213-254
            self.e_embed_list.append(_embed(data.num_edge_features if i == 0 else dim3, dim3))
            self.x_embed_list.append(_embed(self.gc_dim if i == 0 else dim3, dim3))
            self.u_embed_list.append(_embed(data[0].u.shape[1] if i == 0 else dim3, dim3))
            args = (dim3, act, batch_norm, batch_track_stats, dropout_rate, gc_fc_count)
            self.conv_list.append(ops.MetaLayer(MegnetEdgeModel(*args), MegnetNodeModel(*args),
                                                MegnetGlobalModel(*args)))
        self._finish(dim2, post_fc_count, dim3, pool_mult=3, set2set_names=("set2set_x", "set2set_e"))

    def forward(self, data):
        out = self._pre(data)
        x = e = u = None
        for i, conv in enumerate(self.conv_list):
            e_t = self.e_embed_list[i](data.edge_attr if i == 0 else e)
            x_t = self.x_embed_list[i](out if i == 0 else x)
            u_t = self.u_embed_list[i](data.u if i == 0 else u)
            x_o, e_o, u_o = conv(x_t, data.edge_index, e_t, u_t, data.batch)
            if i == 0:  # megnet.py:313-325 — first layer adds the embedded inputs
                x, e, u = x_o + x_t, e_o + e_t, u_o + u_t
            else:       # megnet.py:334-336 — later layers add the running state
                x, e, u = x_o + x, e_o + e, u_o + u
        row = data.edge_index[0]
        if self.pool_order == "early":
            if self.pool == "set2set":
                x_pool = self.set2set_x(x, data.batch)
                e_pool = self.set2set_e(ops.scatter(e, row, 0, None, "mean"), data.batch)
            else:
                x_pool = ops.scatter(x, data.batch, 0, None, self.pool_reduce)
                e_pool = ops.scatter(ops.scatter(e, row, 0, None, self.pool_reduce), data.batch, 0, None,
                                     self.pool_reduce)
            out = self._post(torch.cat([x_pool, e_pool, u], dim=1))
        else:
            out = self._post(x)
            if self.pool == "set2set":
                out = self.lin_out_2(self.set2set_x(out, data.batch))
            else:
                out = ops.POOLS[self.pool](out, data.batch)
        return out.view(-1) if out.shape[1] == 1 else out


REGISTRY = {"CGCNN": CGCNN, "SchNet": SchNet, "MEGNet": MEGNet, "MPNN": MPNN, "GCN": GCN}
