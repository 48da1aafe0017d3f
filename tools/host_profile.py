"""cProfile of the eager forward (host-side launch path) of the fused int8-sim mobilenet1.0."""
import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from quantization.mxnet_amd import mx
net = bench.build_net("mobilenet1.0", 1000, mx.gpu(0))
X = mx.nd.NDArray(torch.randn(128, 3, 224, 224, device="cuda"))
for _ in range(5):
    net(X)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    net(X)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
import time
t0 = time.perf_counter()
for _ in range(50):
    net(X)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step: %.3f ms; incl. drain: %.3f ms" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
