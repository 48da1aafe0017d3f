#!/usr/bin/env python3
"""CPU baseline micro-benchmark: the C++/OpenMP restatement (oracle/libfq_host.so) and the numpy op chain on the headline
tensor / a bounded sample of it, with the core count stated.  Test infrastructure; prints one JSON object."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import host as H  # noqa: E402
from oracle import fq_oracle as O  # noqa: E402


def best_of(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="images of (64,112,112) (128 = the 411 MB headline tensor)")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(7)
    x = np.maximum(rng.standard_normal((args.batch, 64, 112, 112), dtype=np.float32), 0) * np.float32(1.7)
    y = np.empty_like(x)
    tmp = np.empty(2 * x.size, np.float32)
    mb = x.nbytes / 1e6
    out = {"cores": os.cpu_count(), "omp_threads": H.threads(), "tensor_mb": mb}
    for name, fn, bpe in (("host_online_fused", lambda: H.fake_quant_online(x), 12),
                          ("host_unfused_chain", lambda: H.unfused_chain(x, tmp=tmp, out=y), 44),
                          ("host_absmax", lambda: H.absmax_per_sample(x), 4),
                          ("host_histogram", lambda: H.histogram_accumulate(x, 8.0), 4)):
        t = best_of(fn, args.reps)
        out[name] = {"s": t, "alg_gbps": bpe * x.size / t / 1e9, "melems_per_s": x.size / t / 1e6}
    H.set_threads(1)
    t = best_of(lambda: H.fake_quant_online(x[:4]), 1)
    out["host_online_fused_1thread"] = {"s": t, "melems_per_s": x[:4].size / t / 1e6}
    H.set_threads(0)
    t = best_of(lambda: O.unfused_reference_chain(x[:4]), 1)
    out["numpy_unfused_chain_1thread"] = {"s": t, "melems_per_s": x[:4].size / t / 1e6}
    hs = np.abs(rng.standard_normal((8, 2048))).astype(np.float32) * 1000
    t = best_of(lambda: H.kl_search(hs, 256, 256), 1)
    out["host_kl_search_per_layer_s"] = t / 8
    print(json.dumps(out))


if __name__ == "__main__":
    main()
