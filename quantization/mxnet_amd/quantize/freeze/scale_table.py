#-*- coding: utf-8 -*-
"""Scale-table export: what a third-party int8 runtime needs from a calibrated net, instead of
the reference's MKLDNN symbol rewrite (`FreezeHelper`, quantize/freeze/freeze.py:134-238, out of scope).

The reference's deployment note (README.md:267-274, "Generate scales table ... ncnn": weights AND activations int8,
BatchNorm fused into the convolution before the weight scales are taken, per-channel weight quantisation) fixes the
content; the layout follows ncnn's int8 calibration table, one line per tensor, multiplier form (code = round(x * scale)):

    <layer>_param_0 <scale of out-channel 0> <scale of out-channel 1> ...      weight scales, (2^(w-1)-1) / max|w_c|
    <layer> <scale>                                                            activation (layer input) scale

`export_scale_table(net)` reads every converted Conv2D / Dense of a calibrated net: weights as the forward would
quantise them (fake-BN fold applied when the block still folds on the fly; a net processed by `merge_bn` or with
frozen parameters already holds folded / quantised weights), thresholds from `input_max`.  Pure host code; the
per-channel abs-max is a tensor-library reduction on whatever device the parameters live on.
"""
import json

import torch

from ...mx.gluon import nn

__all__ = ["export_scale_table", "format_scale_table"]


def _levels(width, signed):
    return float(2 ** (width - 1) - 1) if signed else float(2 ** width - 1)


def _effective_weight(m):
    """The tensor the forward hands to the weight quantiser (convert_conv2d.py:47-51 fold when fake_bn is live)."""
    w = m.weight.data()._t.detach().float()
    qa = m.quantize_args
    if isinstance(m, nn.Conv2D) and getattr(qa, "fake_bn", False) and getattr(m, "fixed_params", -1) != 1:
        g = m.gamma.data()._t.detach().float()
        var = m.running_var.data()._t.detach().float()
        w = w * (g / torch.sqrt(var + 1e-10)).reshape(-1, *([1] * (w.dim() - 1)))
    return w


def export_scale_table(net, path=None, weight_width=None, input_width=None, per_channel=True, json_path=None):
    """Returns an ordered list of entries {name, kind: 'weight'|'input', scales: [...], threshold(s)}; writes the text
    table to `path` and a JSON rendering to `json_path` when given.  Widths default to each block's own
    `quantize_args`; pass 8 / 8 for an int8 runtime regardless of the simulated widths."""
    entries = []
    for m in net.collect_quantized_blocks():
        if not isinstance(m, (nn.Conv2D, nn.Dense)):
            continue
        qa = m.quantize_args
        ww = int(weight_width or qa.wt_width)
        w = _effective_weight(m)
        rows = w.reshape(w.shape[0], -1)
        if per_channel:
            mx_w = rows.abs().amax(dim=1)
        else:
            mx_w = rows.abs().amax().reshape(1)
        mx_w = mx_w.cpu().double()
        lv = _levels(ww, True)
        scales = [(lv / v) if v > 0 else 0.0 for v in mx_w.tolist()]
        entries.append({"name": m.name + "_param_0", "kind": "weight", "width": ww, "max_abs": mx_w.tolist(),
                        "scales": scales})
    for m in net.collect_quantized_blocks():
        if not isinstance(m, (nn.Conv2D, nn.Dense)) or not m.quantize_args.quantize_input:
            continue
        qa = m.quantize_args
        iw = int(input_width or qa.in_width)
        thr = float(m.input_max.data()._t.detach().reshape(-1)[0].cpu())
        signed = bool(qa.in_signed)
        lv = _levels(iw, signed) if input_width is None else _levels(iw, True)
        entries.append({"name": m.name, "kind": "input", "width": iw, "signed": signed, "threshold": thr,
                        "scales": [(lv / thr) if thr > 0 else 0.0]})
    if path is not None:
        with open(path, "w") as f:
            f.write(format_scale_table(entries))
    if json_path is not None:
        with open(json_path, "w") as f:
            json.dump(entries, f, indent=1)
    return entries


def format_scale_table(entries):
    return "".join("%s %s\n" % (e["name"], " ".join("%.9g" % s for s in e["scales"])) for e in entries)
