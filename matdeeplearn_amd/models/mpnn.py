"""MPNN — /root/reference/matdeeplearn/models/mpnn.py:17-188: NNConv(gc_dim, gc_dim, nn=Seq(Linear(G,dim3),
ReLU, Linear(dim3, gc_dim^2)), aggr="mean") (:83-88) -> BN -> act -> dropout -> one GRU step with
h0 = pre-FC output (:141-161)."""
import torch.nn.functional as F
from torch import nn

from ..nn import NNConv
from ._base import GraphModel


class MPNN(GraphModel):
    def __init__(self, data, dim1=64, dim2=64, dim3=64, pre_fc_count=1, gc_count=3, post_fc_count=1,
                 pool="global_mean_pool", pool_order="early", batch_norm="True", batch_track_stats="True",
                 act="relu", dropout_rate=0.0, compute_dtype="fp32", **kwargs):
        super().__init__()
        self._begin(data, dim1, pre_fc_count, gc_count, pool, pool_order, batch_norm, batch_track_stats, act,
                    dropout_rate, compute_dtype, lists=("conv_list", "gru_list", "bn_list"))
        c = self.gc_dim
        for _ in range(gc_count):
            net = nn.Sequential(nn.Linear(data.num_edge_features, dim3), nn.ReLU(), nn.Linear(dim3, c * c))
            self.conv_list.append(NNConv(c, c, net, aggr="mean"))
            self.gru_list.append(nn.GRU(c, c))
            self._add_bn(c)
        self._finish(dim2, post_fc_count, c)

    def forward(self, data):
        x, edge_attr, csr = self._inputs(data)
        cd = self.compute_dtype
        out = self._pre(x)                  # NNConv (K7) + BatchNorm in the compute dtype; the GRU (library) in fp32
        h = out.float().unsqueeze(0)
        for i, conv in enumerate(self.conv_list):
            m = self._bn(i, conv(out, None, edge_attr, csr=csr))
            m = self._drop(getattr(F, self.act)(m))
            out32, h = self.gru_list[i](m.float().unsqueeze(0), h)
            out = out32.squeeze(0).to(cd)
        return self._head(out, data)
