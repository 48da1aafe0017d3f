import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops
dev = torch.device("cuda", 0)
n = 128
LAYERS = [(32, 64, 112), (64, 128, 56), (128, 128, 56), (128, 256, 28), (256, 256, 28), (256, 512, 14), (512, 512, 14), (512, 1024, 7), (1024, 1024, 7)]
for cin, cout, hw in LAYERS:
    torch.manual_seed(7)
    x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    stat = ops.absmax_per_sample(x)
    cur = torch.empty(1, device=dev)
    codes, scales, rowsum = ops.weight_codes(w, cout, 8)
    for _ in range(4):
        ops.pwconv_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc, bn_shift=sh, act="relu")
torch.cuda.synchronize()
