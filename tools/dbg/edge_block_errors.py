import sys; sys.path.insert(0, "/root/repo")
import tests.test_gpu_workloads as t
import torch
orig = t._close
def loud(a, b, rel, what=""):
    a_, b_ = a.detach().float().cpu(), b.detach().float().cpu()
    s = float(b_.abs().max()) + 1e-12
    print("%-40s err/scale %.4f (allowed %.3f)" % (what, float((a_ - b_).abs().max()) / s, rel))
t._close = loud
t.test_cfg4_megnet_edge_block_with_batchnorm_bf16_gradients()
