"""`convert_model(net, exclude=[], convert_fn=default_convert_fn, custom_fn={})`, `convert_to_relu6(net, exclude=[])`,
`default_convert_fn` — reference API: quantize/convert/convert.py:35-137.

`convert_model` walks the net, applies to every block the converter registered for its EXACT type (`custom_fn[block]`
overrides per instance, `exclude` is matched by identity, :58-63) and gives the net the reference's control methods:
`update_ema(momentum=0.9)`, `collect_quantized_blocks()`, `quantize_input(enable=True, online=True)`,
`enable_quantize()`, `disable_quantize()`, `fix_params()`.  Here they are bound methods of one `NetControls` object
per net (`net._fq_controls`).

MI355X-first difference in `update_ema` (:66-79): the L per-block scalars (`input_max`, `current_input_max`, `act_max`,
`current_act_max`) are re-homed into two contiguous device vectors (an "arena"; each block's Parameter becomes a view),
so one calibration step is ONE `fq_ema_update` launch over L scalars and — across GPUs — ONE small collective
(dist.py), with no host round trip.  The arithmetic per scalar is the reference's: (1-m)*current + m*state in fp32.
"""
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

# block type -> converter (None = leave alone); the commented alternatives are the reference's own
default_convert_fn = {
    nn.Conv2D: gen_conv2d_converter(),
    nn.Dense: gen_dense_converter(),
    nn.Activation: None,     # convert_relu_to_relu6 / gen_act_converter()
    nn.BatchNorm: None,      # bypass_bn
}

_QUANTISABLE = (nn.Dense, nn.Conv2D, nn.Activation)


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


_MODE_EPOCH = [0]


def mode_epoch():
    """Counts the calls that change WHICH kernels a converted net's forward launches or the buffers they read - quantize_input,
    enable / disable, fix_params, (un)fusing - as opposed to the values in those buffers (calibration).  Anything that recorded
    a forward for replay (the CLI's evaluation graphs) is valid for as long as this number stands."""
    return _MODE_EPOCH[0]


def bump_mode_epoch():
    _MODE_EPOCH[0] += 1


class NetControls(object):
    """The control surface `convert_model` installs on a net."""

    # name on the net -> method here
    EXPORTS = {"collect_quantized_blocks": "blocks", "calibration_arena": "arena", "update_ema": "update_ema",
               "quantize_input": "quantize_input", "enable_quantize": "enable", "disable_quantize": "disable",
               "fix_params": "fix_params"}

    def __init__(self, net):
        self.net = net

    def install(self):
        for public, mine in self.EXPORTS.items():
            setattr(self.net, public, getattr(self, mine))
        self.net._fq_controls = self
        return self.net

    # :82-89
    def blocks(self):
        found = []
        self.net.apply(lambda m: found.append(m) if type(m) in _QUANTISABLE and hasattr(m, 'quantize_args') else None)
        return found

    def arena(self):
        """Bind (or re-bind after `reset_ctx`) the contiguous calibration vectors; returns the arena or None."""
        slots = _calibration_slots(self.blocks())
        if not slots:
            return None
        arena = getattr(self.net, "_fq_arena", None)
        if arena is None or not arena.valid(slots):
            device = getattr(slots[0][0], slots[0][1]).data()._t.device
            ops.require_hip(device, "update_ema: calibration state")
            arena = self.net._fq_arena = _Arena(slots, device)
        return arena

    # :66-79
    def update_ema(self, momentum=0.9):
        if ops.in_flight():
            raise RuntimeError("update_ema() inside ops.batches_in_flight(): forwards declared as batches in flight keep "
                               "their batch statistics per stream and do not update current_input_max - calibrate "
                               "outside the declaration (one batch at a time)")
        arena = self.arena()
        if arena is not None:
            sync = getattr(self.net, "_fq_calibration_sync", None)
            if sync is not None:
                sync(self.net, arena)        # multi-GPU: current_* <- statistic of the GLOBAL batch (dist.py)
            ops.ema_update(arena.state, arena.cur, momentum)
        keep, fresh = momentum, 1 - momentum
        for blk in self.blocks():            # fake-BN vectors (not on any BASELINE config): plain NDArray arithmetic
            for moving, batch in (("running_mean", "current_mean"), ("running_var", "current_var")):
                param = getattr(blk, moving, None)
                if param is not None and hasattr(blk, batch):
                    param.set_data(fresh * getattr(blk, batch) + keep * param.data())

    # :92-102
    def quantize_input(self, enable=True, online=True):
        changed = False                      # (the mode epoch moves only when a flag does: the CLI calls this before every
        for blk in self.blocks():            #  calibration epoch and evaluation, and recorded forwards should survive that)
            if type(blk) is nn.Activation:
                if enable and not blk.quantize_args.quantize_act:
                    raise AssertionError("%s was converted with quantize_act=False" % blk.name)
                changed |= (getattr(blk, "quantize_act", None), getattr(blk, "quantize_act_offline", None)) != (enable, not online)
                blk.quantize_act, blk.quantize_act_offline = enable, not online
            elif type(blk) in (nn.Dense, nn.Conv2D):
                if enable and not blk.quantize_args.quantize_input:
                    raise AssertionError("%s was converted with quantize_input=False" % blk.name)
                changed |= (getattr(blk, "quantize_input", None), getattr(blk, "quantize_input_offline", None)) != (enable, not online)
                blk.quantize_input, blk.quantize_input_offline = enable, not online
        if changed:
            bump_mode_epoch()

    # :105-114
    def _switch(self, on):
        changed = False
        for blk in self.blocks():
            changed |= getattr(blk, "enable_quantize", None) != on
            blk.enable_quantize = on
        if changed:
            bump_mode_epoch()

    def enable(self):
        self._switch(True)

    def disable(self):
        self._switch(False)

    # :117-121 — convolutions only, as in the reference (a Dense never freezes)
    def fix_params(self):
        bump_mode_epoch()
        for blk in self.blocks():
            if isinstance(blk, nn.Conv2D):
                blk.fixed_params = 0
                blk.__dict__.pop("_fq_pw_cache", None)      # integer codes of the previous freeze
                blk.__dict__.pop("_fq_no_int8", None)       # ... and its verdict on re-derived codes


def convert_model(net, exclude=[], convert_fn=default_convert_fn, custom_fn={}):
    """
    Convert the model to the one with simulated quantization (in place; the net is also returned).
    :param net: gluon Block to convert.
    :param exclude: list of blocks to leave alone (matched by identity).
    :param convert_fn: dict {block type: func(block) -> None}, applied to blocks of exactly that type.
    :param custom_fn: dict {block instance: func(block) -> None}, overrides `convert_fn` for single blocks.
    """
    def visit(block):
        if any(block is e for e in exclude):
            return
        fn = custom_fn[block] if block in custom_fn else convert_fn.get(type(block))
        if fn is not None:
            fn(block)
    net.apply(visit)
    return NetControls(net).install()


def convert_to_relu6(net, exclude=[]):
    """Every relu Activation of the net (except `exclude`, by identity) becomes a ReLU6 (:124-137)."""
    def visit(block):
        if isinstance(block, nn.Activation) and block._act_type == "relu" and not any(block is e for e in exclude):
            convert_relu_to_relu6(block)
    return net.apply(visit)
