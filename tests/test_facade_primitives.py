"""The primitive semantics the golden vectors G4-G11 rest on, checked WITHOUT the oracle.

The reference's arithmetic lives in Apache MXNet, which is absent from the image: the goldens were made by running
the reference's own files over this package's `mx` facade, so the facade's primitives are part of the parity pin.  Here they are
held against the examples MXNet's own operator documentation prints (python/mxnet/ndarray/gen_op.py docstrings of round, rint-free
path, clip, cast, abs, sum, max: the numbers below are those documented vectors), and the one place where the reference's result
depends on the numpy version of the machine that runs it is enumerated."""
import numpy as np

from quantization.mxnet_amd import mx

nd = mx.nd


def A(x, dtype=np.float32):
    return nd.array(np.asarray(x, dtype=dtype))


def test_round_is_half_away_from_zero_as_documented():
    # mx.nd.round: "round([-1.5, 1.5, -1.9, 1.9, 2.1]) = [-2., 2., -2., 2., 2.]"
    assert nd.round(A([-1.5, 1.5, -1.9, 1.9, 2.1])).asnumpy().tolist() == [-2.0, 2.0, -2.0, 2.0, 2.0]
    # ... and the ties numpy's round-half-even would get wrong, the largest fp32 below 0.5 (must not round up), -0.0
    got = A([0.5, 2.5, -2.5, 0.49999997, -0.49999997, 8388609.0, -0.0]).round().asnumpy()
    assert got.tolist() == [1.0, 3.0, -3.0, 0.0, -0.0, 8388609.0, -0.0]
    assert np.signbit(got[4]) and np.signbit(got[6])


def test_clip_as_documented():
    # mx.nd.clip: "x = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]; clip(x, 1, 8) = [1, 1, 2, 3, 4, 5, 6, 7, 8, 8]"
    assert nd.clip(A(np.arange(10)), 1, 8).asnumpy().tolist() == [1, 1, 2, 3, 4, 5, 6, 7, 8, 8]
    assert A([-3.0, 0.25, 9.0]).clip(0.0, 1.5).asnumpy().tolist() == [0.0, 0.25, 1.5]


def test_cast_truncates_toward_zero_as_documented():
    # mx.nd.cast: "cast([0.9, 1.3], dtype='int32') = [0, 1]"
    assert nd.cast(A([0.9, 1.3]), dtype="int32").asnumpy().tolist() == [0, 1]
    assert nd.cast(A([-0.9, -1.7, 126.99, -127.5]), dtype="int32").asnumpy().tolist() == [0, -1, 126, -127]
    assert nd.cast(A([0.0, 255.0, 17.9]), dtype="uint8").asnumpy().tolist() == [0, 255, 17]


def test_abs_sum_max_as_documented():
    # mx.nd.abs: "abs([-2, 0, 3]) = [2, 0, 3]"
    assert nd.abs(A([-2, 0, 3])).asnumpy().tolist() == [2, 0, 3]
    # mx.nd.sum / mx.nd.max: data = [[[1,2],[2,3],[1,3]], [[1,4],[4,3],[5,2]], [[7,1],[7,2],[7,3]]]
    data = A([[[1, 2], [2, 3], [1, 3]], [[1, 4], [4, 3], [5, 2]], [[7, 1], [7, 2], [7, 3]]])
    assert nd.sum(data, axis=1).asnumpy().tolist() == [[4, 8], [10, 9], [21, 6]]             # "sum(data, axis=1)"
    assert nd.sum(data, axis=(1, 2)).asnumpy().tolist() == [12, 19, 27]                      # "sum(data, axis=[1,2])"
    assert nd.max(data, axis=1).asnumpy().tolist() == [[2, 3], [5, 4], [7, 3]]
    assert nd.max(data, axis=(1, 2)).asnumpy().tolist() == [3, 5, 7]
    assert float(data.abs().max(axis=(1, 2)).mean().asscalar()) == 5.0                      # convert_conv2d.py:56's chain


def test_tensor_by_scalar_is_the_ieee_fp32_division_by_the_fp32_scalar():
    # MXNet's _div_scalar: the python scalar becomes DType(scalar) and every element is divided by it (no reciprocal trick)
    x = np.float32([1.0, 0.3, 7.0, 1e-3, 123.456])
    for d in (3.0, 0.1, 1.0 / 255.0 + 1e-10, 7e-4):
        want = x / np.float32(d)
        assert np.array_equal((A(x) / d).asnumpy(), want)


def test_where_the_epsilon_add_depends_on_the_numpy_version_of_the_reference_machine():
    """ste_func.py:39-41 divides by `self.scale + 1e-10`.  For activations `scale` is a numpy fp32 SCALAR (convert_conv2d.py:56-64:
    `.asscalar()` of an fp32 NDArray, divided by a python int), so the sum is numpy's: fp64 under numpy 1.x (scalar + python float
    promotes), fp32 under numpy 2 (NEP 50) - MXNet then casts whatever it gets to fp32 for `_div_scalar`.  The two divisors
        d64 = fp32(fp64(s) + 1e-10)        (numpy 1.x: one rounding of the exact sum)
        d32 = fp32(s) + fp32(1e-10)        (numpy 2, and per-channel weight scales in either: an fp32 `_plus_scalar`)
    differ only where the 1.3e-18 by which fp32(1e-10) exceeds 1e-10 moves the sum across an fp32 rounding boundary.  This
    build follows d32 (the oracle, the kernels' make_qparams, and the goldens - generated under numpy 2 - agree on it); the
    enumeration below records how rarely that choice can matter: NEVER for a scale of 1e-8 or more among 16 million scales
    swept down from 1e3 (thresholds from 2.5e-6 up at 8 bits), and for about two scales in a hundred below 1e-8, where 1e-10
    is itself a sizeable part of the divisor - by one ulp of the divisor, which flips a code only for inputs that sit within
    2^-24 (relative) of a rounding tie."""
    rng = np.random.default_rng(0)
    levels = np.float32(255.0)
    for lo, hi, bound in ((1e-3, 1e3, 0.0), (1e-5, 1e-3, 0.0), (2e-7, 1e-5, 0.0), (1e-8, 2e-7, 0.0), (1e-10, 1e-8, 3e-2)):
        thr = np.exp(rng.uniform(np.log(lo), np.log(hi), 4_000_000)).astype(np.float32) * levels
        s = (thr / levels).astype(np.float32)
        d64 = (s.astype(np.float64) + 1e-10).astype(np.float32)
        d32 = s + np.float32(1e-10)
        differ = d64 != d32
        assert differ.mean() <= bound, (lo, hi, float(differ.mean()))
        if differ.any():                                         # never by more than one ulp
            i = np.flatnonzero(differ)
            assert np.all(np.abs(d64[i].view(np.int32).astype(np.int64) - d32[i].view(np.int32)) == 1)
    # the thresholds the five BASELINE nets produce on the synthetic inputs (1e-2 ... 3e1) lie in the
    # first range: both numpy versions give the same divisor, hence the same codes
