import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops
dev = torch.device("cuda", 0)
n, c, hw, s = 128, 32, 112, 1
torch.manual_seed(7)
x = torch.relu(torch.randn(n, c, hw, hw, device=dev))
w = torch.randn(c, 1, 3, 3, device=dev) * 0.3
sc = torch.rand(c, device=dev) + 0.5
sh = torch.randn(c, device=dev)
stat = ops.absmax_per_sample(x)
cur = torch.empty(1, device=dev)
for _ in range(3):
    ops.dwconv3x3(x, w, None, stride=s, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc, bn_shift=sh, act="relu")
    ops.dwconv3x3(x, w, None, stride=s, bn_scale=sc, bn_shift=sh, act="relu")
torch.cuda.synchronize()
