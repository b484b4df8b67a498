"""SchNet — /root/reference/matdeeplearn/models/schnet.py:16-172: conv_list.i = InteractionBlock(gc_dim,
num_edge_features, dim3, cutoff) (:81); out = out + conv(out, edge_index, edge_weight, edge_attr) then
BatchNorm (:134-143), no activation between layers."""
from torch import nn

from .. import ops
from ..nn import InteractionBlock, cosine_cutoff
from ._base import GraphModel


class SchNet(GraphModel):
    def __init__(self, data, dim1=64, dim2=64, dim3=64, cutoff=8, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, compute_dtype="fp32", **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, compute_dtype)
        for _ in range(gc_count):
            self.conv_list.append(InteractionBlock(self.gc_dim, data.num_edge_features, dim3, cutoff))
            self._add_bn(self.gc_dim)
        self._finish(dim2, post_fc_count, self.gc_dim)

    def forward(self, data):
        x, edge_attr, csr = self._inputs(data)
        out = self._pre(x)
        cut = cosine_cutoff(data.edge_weight, self.conv_list[0].conv.cutoff) if len(self.conv_list) else None   # once per batch
        by_source = ops.BySourceAttrs()                 # the backward's by-source copy of (edge_attr, cut): made once, by the last block
        for i, conv in enumerate(self.conv_list):
            out = self._drop(self._bn(i, out + conv(out, None, data.edge_weight, edge_attr, csr=csr, cut=cut, by_source=by_source)))
        return self._head(out, data)
