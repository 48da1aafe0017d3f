#-*- coding: utf-8 -*-
"""quantize.freeze — `merge_bn` and the scale-table export (`scale_table.py`) are provided.

The reference's `FreezeHelper` / `quantize_symbol` (quantize/freeze/freeze.py) drive libmxnet's Intel-MKLDNN subgraph
quantiser through `_LIB.MXQuantizeSymbol`; that is a graph pass of a third-party binary for another vendor's CPU
backend, marked untested by the reference's README, and is out of scope here."""
from .merge_bn import *
from .scale_table import export_scale_table, format_scale_table


def __getattr__(name):
    if name in ("FreezeHelper", "quantize_symbol", "quantize_params", "calibrate_quantized_sym"):
        raise NotImplementedError(
            "%s exports to MXNet's MKLDNN int8 symbol format via libmxnet; not part of the MI355X fake-quant path "
            "(see DESIGN.md, out of scope)" % name)
    raise AttributeError(name)
