"""`convert_model`, `convert_to_relu6`, `default_convert_fn` — reference: quantize/convert/convert.py:35-137.

`convert_model(net, exclude, convert_fn, custom_fn)` walks the net with `net.apply`, dispatches on the EXACT block type
(`convert_fn.get(type(m))`, per-instance override `custom_fn[m]`, identity-based `exclude`, :58-63) and injects the same
control methods on the net: `update_ema`, `collect_quantized_blocks`, `quantize_input`, `enable_quantize`,
`disable_quantize`, `fix_params`.

MI355X-first difference in `update_ema` (:66-79): the L per-block scalars (`input_max`, `current_input_max`, `act_max`,
`current_act_max`) are re-homed into two contiguous device vectors (an "arena"; each block's Parameter becomes a view),
so one calibration step is ONE `fq_ema_update` launch over L scalars and — across GPUs — ONE small collective
(dist.py), with no host round trip.  The arithmetic per scalar is the reference's: (1-m)*current + m*state in fp32.
"""
import types

import torch

from ...mx.gluon import nn
from ...mx.ndarray import NDArray
from ... import ops
from .._state import DeviceScalar
from .convert_conv2d import gen_conv2d_converter
from .convert_dense import gen_dense_converter
from .convert_act import gen_act_converter, convert_relu_to_relu6
from .convert_bn import bypass_bn

__all__ = ["convert_model", "convert_to_relu6", 'default_convert_fn']

default_convert_fn = {
    nn.Conv2D: gen_conv2d_converter(),
    nn.Dense: gen_dense_converter(),
    nn.Activation: None,  # convert_relu_to_relu6,  # gen_act_converter(),
    nn.BatchNorm: None  # bypass_bn
}


class _Arena(object):
    """Contiguous homes for the per-block calibration scalars of one net."""

    def __init__(self, slots, device):
        self.slots = slots                       # [(block, param_attr, cur_attr, public_cur_attr)]
        n = len(slots)
        self.state = torch.zeros(n, dtype=torch.float32, device=device)
        self.cur = torch.zeros(n, dtype=torch.float32, device=device)
        for i, (blk, pattr, cattr, pub) in enumerate(slots):
            param = getattr(blk, pattr)
            self.state[i:i + 1].copy_(param.data()._t.reshape(1))
            param._data = NDArray(self.state[i:i + 1])
            old = getattr(blk, cattr, None)
            if isinstance(old, torch.Tensor):
                self.cur[i:i + 1].copy_(old.reshape(1))
            setattr(blk, cattr, self.cur[i:i + 1])
            setattr(blk, pub, DeviceScalar(self.cur[i:i + 1]))

    def valid(self, slots):
        if len(slots) != len(self.slots):
            return False
        sp, cp = self.state.data_ptr(), self.cur.data_ptr()
        for i, ((blk, pattr, cattr, _), (blk0, pattr0, _, _)) in enumerate(zip(slots, self.slots)):
            if blk is not blk0 or pattr != pattr0:
                return False
            p = getattr(blk, pattr)._data
            c = getattr(blk, cattr, None)
            if p is None or p._t.data_ptr() != sp + 4 * i or not isinstance(c, torch.Tensor) \
                    or c.data_ptr() != cp + 4 * i:
                return False
        return True


def _calibration_slots(blocks):
    slots = []
    for b in blocks:
        if getattr(b, "input_max", None) is not None:
            slots.append((b, "input_max", "_fq_cur", "current_input_max"))
        if getattr(b, "act_max", None) is not None:
            slots.append((b, "act_max", "_fq_cur", "current_act_max"))
    return slots


def convert_model(net, exclude=[], convert_fn=default_convert_fn, custom_fn={}):
    """
    Convert the model to the one with simulated quantization.
    :param net: gluon Block
        The net to convert.
    :param exclude: list of Block
        Blocks that want to exclude.
    :param convert_fn: dict with (module, func) key-value pairs
        `module`: Block type; `func`: function `func(module) -> None` applied to blocks of exactly that type.
    :param custom_fn: dict with (block instance, func) pairs overriding `convert_fn` for single blocks.
    :return: the converted net (converted in place).
    """
    # Convert network (:58-63)
    def _convert(m):
        if not any(m is e for e in exclude):
            fn = custom_fn[m] if m in custom_fn else convert_fn.get(type(m))
            if fn is not None:
                fn(m)
    net.apply(_convert)

    # Add a method to collect all quantized blocks (:82-89)
    def _collect_quantized_blocks(self):
        blocks = []

        def _collect_blocks(m):
            if type(m) in (nn.Dense, nn.Conv2D, nn.Activation) and hasattr(m, 'quantize_args'):
                blocks.append(m)
        net.apply(_collect_blocks)
        return blocks
    net.collect_quantized_blocks = types.MethodType(_collect_quantized_blocks, net)

    def _calibration_arena(self):
        """Bind (or re-bind after `reset_ctx`) the contiguous calibration vectors; returns the arena or None."""
        slots = _calibration_slots(self.collect_quantized_blocks())
        if not slots:
            return None
        arena = getattr(self, "_fq_arena", None)
        if arena is None or not arena.valid(slots):
            device = getattr(slots[0][0], slots[0][1]).data()._t.device
            ops.require_hip(device, "update_ema: calibration state")
            arena = _Arena(slots, device)
            self._fq_arena = arena
        return arena
    net.calibration_arena = types.MethodType(_calibration_arena, net)

    # Add method to update ema for `input_max` / `act_max` (and fake-bn statistics) (:66-79)
    def _update_ema(self, momentum=0.9):
        arena = self.calibration_arena()
        if arena is not None:
            sync = getattr(self, "_fq_calibration_sync", None)
            if sync is not None:
                sync(self, arena)        # multi-GPU: current_* <- statistic of the GLOBAL batch (dist.py)
            ops.ema_update(arena.state, arena.cur, momentum)
        for qblocks in self.collect_quantized_blocks():
            # if fake bn (vectors; not on any BASELINE config): the reference's own expression on NDArrays
            if getattr(qblocks, "running_mean", None) is not None and hasattr(qblocks, "current_mean"):
                qblocks.running_mean.set_data((1 - momentum) * qblocks.current_mean + momentum * qblocks.running_mean.data())
            if getattr(qblocks, "running_var", None) is not None and hasattr(qblocks, "current_var"):
                qblocks.running_var.set_data((1 - momentum) * qblocks.current_var + momentum * qblocks.running_var.data())
    net.update_ema = types.MethodType(_update_ema, net)

    # Add method to control the mode of input quantization as online or offline (:92-102)
    def _quantize_input(self, enable=True, online=True):
        for qblocks in self.collect_quantized_blocks():
            if type(qblocks) in (nn.Dense, nn.Conv2D):
                assert (not enable) or qblocks.quantize_args.quantize_input
                qblocks.quantize_input = enable
                qblocks.quantize_input_offline = not online
            elif type(qblocks) == nn.Activation:
                assert (not enable) or qblocks.quantize_args.quantize_act
                qblocks.quantize_act = enable
                qblocks.quantize_act_offline = not online
    net.quantize_input = types.MethodType(_quantize_input, net)

    # Add method to control enable/disable quantization (:105-114)
    def _enable_quantize(self):
        for qblocks in self.collect_quantized_blocks():
            if type(qblocks) in (nn.Dense, nn.Conv2D, nn.Activation):
                qblocks.enable_quantize = True

    def _disable_quantize(self):
        for qblocks in self.collect_quantized_blocks():
            if type(qblocks) in (nn.Dense, nn.Conv2D, nn.Activation):
                qblocks.enable_quantize = False
    net.enable_quantize = types.MethodType(_enable_quantize, net)
    net.disable_quantize = types.MethodType(_disable_quantize, net)

    # Add method to fixed parameters(weights and bias) (:117-121) — Conv2D only, as in the reference
    def _fix_params(self):
        for m in net.collect_quantized_blocks():
            if isinstance(m, nn.Conv2D):
                m.fixed_params = 0
    net.fix_params = types.MethodType(_fix_params, net)

    return net


def convert_to_relu6(net, exclude=[]):
    """Convert ReLUs in net to ReLU6 (:124-137)."""
    def _convert_to_relu6(m):
        if isinstance(m, nn.Activation) and m._act_type == "relu" and not any(m is e for e in exclude):
            convert_relu_to_relu6(m)
    return net.apply(_convert_to_relu6)
