"""GCN — /root/reference/matdeeplearn/models/gcn.py:17-173: GCNConv(gc_dim, gc_dim, improved=True,
add_self_loops=False) called with edge_weight = raw distance (:80-82,135-144) -> BN -> act -> dropout."""
import torch.nn.functional as F
from torch import nn

from ..nn import GCNConv
from ._base import GraphModel


class GCN(GraphModel):
    def __init__(self, data, dim1=64, dim2=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, compute_dtype="fp32", **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, compute_dtype)
        for _ in range(gc_count):
            self.conv_list.append(GCNConv(self.gc_dim, self.gc_dim, improved=True, add_self_loops=False))
            self._add_bn(self.gc_dim)
        self._finish(dim2, post_fc_count, self.gc_dim)

    def forward(self, data):
        x, _, csr = self._inputs(data)
        out = self._pre(x)
        for i, conv in enumerate(self.conv_list):
            out = self._bn(i, conv(out, None, data.edge_weight, csr=csr))
            out = self._drop(getattr(F, self.act)(out))
        return self._head(out, data)
