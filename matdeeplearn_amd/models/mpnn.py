"""MPNN — /root/reference/matdeeplearn/models/mpnn.py:17-188: NNConv(gc_dim, gc_dim, nn=Seq(Linear(G,dim3),
ReLU, Linear(dim3, gc_dim^2)), aggr="mean") (:83-88) -> BN -> act -> dropout -> one GRU step with
h0 = pre-FC output (:141-161)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..nn import NNConv
from ._base import GraphModel


def gru_step(gru, x, h):
    """One step of a single-layer torch.nn.GRU (`out, h = gru(x[None], h[None])`, mpnn.py:160-161) written out: two dense
    products in the dtype of x, the gate arithmetic in fp32 (torch's gate order r | z | n; n uses r * (W_hn h + b_hn)).
    The library call goes through MIOpen's RNN path, which for sequence length 1 spends ~7 ms per layer on 6e4 rows in
    fp32 GEMMs and tensor-op kernels — a third of the MPNN step; the parameters stay those of the nn.GRU module
    (state_dict keys gru_list.{i}.weight_ih_l0, ...).  Returns (h_new fp32, h_new in the dtype of x): on a HIP device the gates are
    one launch per direction (ops.gru_gates, csrc/gru.hip) instead of ~12 + ~25 elementwise / chunk / cat / cast launches."""
    cd = x.dtype
    if cd == torch.bfloat16 and x.is_cuda:
        # (ops.linear: the gate matrices' weight gradients — [3C, C] with the contraction over 6e4 rows — on the TN GEMM)
        gi = ops.linear(x, gru.weight_ih_l0, gru.bias_ih_l0)
        gh = ops.linear(h.to(cd), gru.weight_hh_l0, gru.bias_hh_l0)
    else:
        gi = F.linear(x, gru.weight_ih_l0.to(cd), gru.bias_ih_l0.to(cd))
        gh = F.linear(h.to(cd), gru.weight_hh_l0.to(cd), gru.bias_hh_l0.to(cd))
    if ops.gru_gates_ok(gi, gh, h):
        return ops.gru_gates(gi, gh, h)
    gi, gh = gi.float(), gh.float()
    i_r, i_z, i_n = gi.chunk(3, dim=1)
    h_r, h_z, h_n = gh.chunk(3, dim=1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    h_new = n + z * (h - n)                                   # (1 - z) * n + z * h
    return h_new, h_new.to(cd)


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
        out = self._pre(x)                  # NNConv (K7) + BatchNorm in the compute dtype; the GRU state in fp32
        h = out.float()
        for i, conv in enumerate(self.conv_list):
            m = self._bn(i, conv(out, None, edge_attr, csr=csr))
            m = self._drop(getattr(F, self.act)(m))
            gru = self.gru_list[i]
            if gru.num_layers == 1 and not gru.bidirectional and gru.bias:
                h, out = gru_step(gru, m, h)
            else:
                h = gru(m.float().unsqueeze(0), h.unsqueeze(0))[1].squeeze(0)
                out = h.to(cd)
        return self._head(out, data)
