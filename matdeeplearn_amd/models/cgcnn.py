"""CGCNN model wrapper on the HIP message-passing engine.

Same constructor keywords (string booleans included), forward contract and state_dict keys as
/root/reference/matdeeplearn/models/cgcnn.py:17-174:
    pre_lin_list.{i} -> [conv_list.{i} (CGConv, aggr=mean) -> bn_list.{i} -> dropout] x gc_count
    -> pool (early|late; global_{mean,add,max}_pool) -> post_lin_list.{i} -> lin_out
There is NO activation between conv layers (reference :146 has it commented out).

Extra (non-reference) keyword: compute_dtype = "fp32" (parity mode, default) | "bf16" (storage
of node/edge features in bf16, fp32 accumulation; master weights stay fp32).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..nn import CGConv


class CGCNN(nn.Module):
    def __init__(self, data, dim1=64, dim2=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, compute_dtype="fp32", **kwargs):
        super().__init__()
        assert gc_count > 0, "Need at least 1 GC layer"
        if pool == "set2set":
            raise ops.MdlError("set2set pooling is a later scope row (SURVEY 8f N4); use global_*_pool")
        self.batch_track_stats = batch_track_stats != "False"
        self.batch_norm, self.pool, self.act = batch_norm, pool, act
        self.pool_order, self.dropout_rate = pool_order, dropout_rate
        self.compute_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[compute_dtype]
        gc_dim = data.num_features if pre_fc_count == 0 else dim1
        y0 = data[0].y
        output_dim = 1 if y0.ndim == 0 else len(y0[0])

        self.pre_lin_list = nn.ModuleList(
            [nn.Linear(data.num_features if i == 0 else dim1, dim1) for i in range(pre_fc_count)])
        self.conv_list = nn.ModuleList(
            [CGConv(gc_dim, data.num_edge_features, aggr="mean", batch_norm=False) for _ in range(gc_count)])
        self.bn_list = nn.ModuleList(
            [nn.BatchNorm1d(gc_dim, track_running_stats=self.batch_track_stats) for _ in range(gc_count)]
            if batch_norm == "True" else [])
        self.post_lin_list = nn.ModuleList(
            [nn.Linear(gc_dim if i == 0 else dim2, dim2) for i in range(post_fc_count)])
        self.lin_out = nn.Linear(dim2 if post_fc_count > 0 else gc_dim, output_dim)

    def _dense(self, lin, h):
        """Linear in the compute dtype with fp32 master weights (library GEMM)."""
        if h.dtype == torch.float32:
            return lin(h)
        return F.linear(h, lin.weight.to(h.dtype), lin.bias.to(h.dtype))

    def forward(self, data):
        cd = self.compute_dtype
        out = data.x.to(cd)
        edge_attr = data.edge_attr.to(cd)
        csr = getattr(data, "csr", None)
        if csr is None:
            csr = ops.csr_for(data.edge_index, out.shape[0])
        for lin in self.pre_lin_list:
            out = getattr(F, self.act)(self._dense(lin, out))
        for i, conv in enumerate(self.conv_list):
            out = conv(out, data.edge_index, edge_attr, csr=csr)
            if self.batch_norm == "True":
                out = self.bn_list[i](out)
            out = F.dropout(out, p=self.dropout_rate, training=self.training)
        num_graphs = getattr(data, "num_graphs", None)
        if self.pool_order == "early":
            out = ops.POOLS[self.pool](out, data.batch, num_graphs)
            for lin in self.post_lin_list:
                out = getattr(F, self.act)(self._dense(lin, out))
            out = self._dense(self.lin_out, out)
        else:
            for lin in self.post_lin_list:
                out = getattr(F, self.act)(self._dense(lin, out))
            out = self._dense(self.lin_out, out)
            out = ops.POOLS[self.pool](out, data.batch, num_graphs)
        out = out.float()
        return out.view(-1) if out.shape[1] == 1 else out
