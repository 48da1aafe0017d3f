#-*- coding: utf-8 -*-
"""The reference's `quantize` package surface (quantize/__init__.py:3-9), L2 bodies re-routed to the HIP library.

`freeze` (the libmxnet/MKLDNN symbol exporter) is out of scope — see DESIGN.md; `merge_bn` lives under
`quantize.freeze` as in the reference."""
from . import convert

from . import initialize

from . import freeze

from . import distribution_calibrate

from .utils import *
