"""`collect_qparams(net)` / `print_all_qparams(net)` — reference API: quantize/utils.py:30-52.
The calibrated ranges are the Parameters named `*_min` / `*_max` (every quantised block's `input_max`, an activation's
`act_max`), reported in net order."""
from collections import OrderedDict

__all__ = ['collect_qparams', 'print_all_qparams']

_RANGE_SUFFIXES = ("_min", "_max")


def collect_qparams(net):
    return OrderedDict((name, p) for name, p in net.collect_params().items() if name.endswith(_RANGE_SUFFIXES))


def print_all_qparams(net):
    for name, p in collect_qparams(net).items():
        print("{}:\t\t{:+.4f}".format(name, p.data().asscalar()))
