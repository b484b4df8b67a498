"""debug: bf16 forward error of SchNet (eval) against the HIP fp32 forward, with the lin2 -> ssp -> lin chain fused / unfused"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from matdeeplearn_amd import models, nn as mnn, ops
from matdeeplearn_amd.process import synthetic_mof
from matdeeplearn_amd.training import make_optimizer
dev = torch.device("cuda:0")
ds = synthetic_mof(400, seed=0).to(dev)
kw = dict(dim1=100, dim2=100, dim3=150, cutoff=8, pre_fc_count=1, gc_count=4, post_fc_count=3)
torch.manual_seed(0)
m16 = models.SchNet(ds, compute_dtype="bf16", **kw).to(dev)
# a few training steps so that BatchNorm's running statistics and the weights are not the initial ones
opt = make_optimizer(m16.parameters(), "AdamW", lr=0.002)
m16.train()
for k in range(int(os.environ.get("STEPS", 20))):
    b = ds.collate(np.arange(k * 16 % 256, k * 16 % 256 + 64), edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    opt.zero_grad()
    loss = torch.nn.functional.l1_loss(m16(b), b.y)
    loss.backward()
    opt.step()
m32 = models.SchNet(ds, compute_dtype="fp32", **kw).to(dev)
m32.load_state_dict(m16.state_dict())
m16.eval(); m32.eval()
ids = np.arange(300, 364)
with torch.no_grad():
    p32 = m32(ds.collate(ids, edge_dtype=torch.float32, x_dtype=torch.float32))
    b16 = ds.collate(ids, edge_dtype=torch.bfloat16, x_dtype=torch.bfloat16)
    orig = mnn.InteractionBlock.forward

    def unchained(self, x, edge_index, edge_weight, edge_attr, csr=None, cut=None, by_source=None):
        return mnn._lin(self.lin, self.act(self.conv(x, edge_index, edge_weight, edge_attr, csr=csr, cut=cut, by_source=by_source)))
    for name, f in (("chained", orig), ("unchained", unchained)):
        mnn.InteractionBlock.forward = f
        for fused in (True, False):
            ops.configure(cfconv_fused=fused)
            p16 = m16(b16)
            d = (p16 - p32)
            print("%-10s cfconv_fused=%d  scale %.3f  max|d| %.4f  mean d %+.4f  mean|d| %.4f  MAE32 %.4f MAE16 %.4f" % (
                name, fused, float(p32.abs().max()), float(d.abs().max()), float(d.mean()), float(d.abs().mean()),
                float((p32 - b16.y).abs().mean()), float((p16 - b16.y).abs().mean())))
