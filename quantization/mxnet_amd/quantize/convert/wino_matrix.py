"""Winograd weight-transform matrices G for F(2,3), F(4,3), F(6,3) as NDArrays — `Winograd_G[variant]`
(reference: quantize/convert/wino_matrix.py:29-60).  The numeric tables live next to the kernel binding
(`ops.winograd_matrices`), which also caches the host-side pseudo-inverses the reference recomputes per call."""
from ...mx import nd
from ... import ops

__all__ = ['Winograd_G']

Winograd_G = {name: nd.array(ops.winograd_matrices(name)[0]) for name in ("F23", "F43", "F63")}
