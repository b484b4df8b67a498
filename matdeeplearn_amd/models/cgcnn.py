"""CGCNN on the HIP message-passing engine — /root/reference/matdeeplearn/models/cgcnn.py:17-174.
conv_list.i = CGConv(gc_dim, num_edge_features, aggr="mean", batch_norm=False) (:80-83); per layer
conv -> bn_list.i -> dropout, NO activation between layers (:146 is commented out in the reference)."""
import torch
from torch import nn

from .. import ops
from ..nn import CGConv
from ._base import GraphModel


class CGCNN(GraphModel):
    # nothing in this model walks the edges by SOURCE (the CGConv backward reaches the source rows from the by-target walk):
    # a static batch built for it skips the by-source CSR (process.StaticBatch(by_source=False), training.GraphedStep)
    needs_by_source = False

    def __init__(self, data, dim1=64, dim2=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, compute_dtype="fp32", **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, compute_dtype)
        for _ in range(gc_count):
            self.conv_list.append(CGConv(self.gc_dim, data.num_edge_features, aggr="mean", batch_norm=False))
            self._add_bn(self.gc_dim)
        self._finish(dim2, post_fc_count, self.gc_dim)

    def forward(self, data):
        x, edge_attr, csr = self._inputs(data)
        out = self._pre(x)
        bn_on = self.batch_norm == "True"
        # every layer's weights are packed for the kernels in ONE launch (they are all known here): at the reference's batch size a
        # pack launch per layer is 5 us of a launch-bound step
        packs = ops.cgconv_prepack(list(self.conv_list), out.dtype, out.device, want_node=torch.is_grad_enabled()) if out.is_cuda else None
        for i, conv in enumerate(self.conv_list):
            # conv -> bn as one call: the BatchNorm's statistics are formed in the conv kernel's epilogue where the layer has the
            # shape for it (nn.CGConv.forward); the sums' shift = the beta of the BatchNorm whose output this layer reads
            prev = self.bn_list[i - 1].bias if (bn_on and i > 0) else None
            out = self._drop(conv(out, None, edge_attr, csr=csr, bn=self.bn_list[i] if bn_on else None, bn_shift=prev,
                                  packed=None if packs is None else packs[i], split=self.split_products))
        return self._head(out, data)
