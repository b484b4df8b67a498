"""Convolution operator modules with the PyG constructor / forward surface the reference uses
(SURVEY.md 8b level b2), backed by the HIP kernels in libmdl_hip.so.

  CGConv(channels, dim, aggr, batch_norm)   matdeeplearn/models/cgcnn.py:80-83 / :136-145
Sub-module and parameter names follow PyG so `state_dict()` keys interchange with the reference
(SURVEY Appendix A.7): CGConv has `lin_f`, `lin_s` = Linear(2*channels + dim, channels).
"""
from torch import nn

from . import ops


class CGConv(nn.Module):
    """out_i = x_i + aggr_{j->i} sigmoid(lin_f z_ij) * softplus(lin_s z_ij), z_ij = [x_i | x_j | e_ij]."""

    def __init__(self, channels, dim=0, aggr="add", batch_norm=False, bias=True, **kwargs):
        super().__init__()
        if isinstance(channels, (tuple, list)):
            if channels[0] != channels[1]:
                raise ops.MdlError("CGConv: bipartite channel pairs are not on the reference path")
            channels = channels[0]
        if batch_norm:
            raise ops.MdlError("CGConv(batch_norm=True) is not used by the reference (cgcnn.py:81)")
        self.channels, self.dim, self.aggr = channels, dim, aggr
        # same construction order / default init as PyG (torch.nn.Linear kaiming-uniform)
        self.lin_f = nn.Linear(2 * channels + dim, channels, bias=bias)
        self.lin_s = nn.Linear(2 * channels + dim, channels, bias=bias)

    def reset_parameters(self):
        self.lin_f.reset_parameters()
        self.lin_s.reset_parameters()

    def forward(self, x, edge_index, edge_attr=None, csr=None):
        if edge_attr is None:
            edge_attr = x.new_zeros((edge_index.shape[1], 0))
        return ops.cgconv(x, edge_index, edge_attr, self.lin_f.weight, self.lin_f.bias, self.lin_s.weight,
                          self.lin_s.bias, self.aggr, csr=csr)

    def extra_repr(self):
        return "%d, dim=%d, aggr=%s" % (self.channels, self.dim, self.aggr)
