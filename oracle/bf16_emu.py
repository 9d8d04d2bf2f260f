"""ORACLE (test infrastructure, NOT product code): bf16-rounding emulation on top of torch_ref.

The bf16 path (BASELINE configs 4-5) stores every activation and activation-gradient tensor in
bf16 and feeds bf16 operands to the MFMA with fp32 accumulation; BN statistics, weight gradients
and master weights stay fp32.  Against the fp32 oracle that path can only be checked loosely
(rounding flips ReLU masks of near-zero activations and small-batch BN amplifies it), so this module
makes the torch-CPU restatement round at the SAME points:
  * conv: operands x and w rounded to bf16, fp32 accumulate, output rounded; dgrad output rounded;
    the weight gradient is NOT rounded (fp32 output of the wgrad kernel)
  * BN (+ReLU) outputs rounded once after the activation; the last BN of a residual block is not
    rounded before the add (the HIP kernel fuses BN + add + ReLU and rounds once)
  * max-pool / block outputs rounded (no-ops on already-bf16 values; they round the GRADIENT)
Gradient rounding happens at the gradient of each rounded tensor.  Agreement with the HIP path is
expected to ~1e-2 (not bitwise: multi-consumer gradients are summed in a different rounding order).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _r(t):
    return t.bfloat16().float()


class RoundBoth(torch.autograd.Function):
    """bf16 rounding of the value (forward) and of its gradient (backward)."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class RoundFwd(torch.autograd.Function):
    """bf16 rounding of the value only (conv weights: the master gradient stays fp32)."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


def emulate_bf16(model, round_weights=lambda conv: conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0):
    """Instruments a torch_ref model in place.  `round_weights(conv)` says whether that conv's
    weights are bf16 operands (the MFMA implicit-GEMM path) or stay fp32 (the direct path used for
    Cin=1/3-style layers)."""
    from . import torch_ref as tr
    no_round_bn = set()
    for m in model.modules():
        if isinstance(m, tr.RefResBottleneck):
            no_round_bn.add(m.res_bottleneck[2][1])
        elif isinstance(m, tr.RefResBlock):
            no_round_bn.add(m.res_block[1][1])
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            def fwd(x, m=m):
                # depthwise kernels read the fp32 master weights directly: never rounded
                w = RoundFwd.apply(m.weight) if (m.groups == 1 and round_weights(m)) else m.weight
                return RoundBoth.apply(F.conv2d(x, w, m.bias, m.stride, m.padding, m.dilation, m.groups))
            m.forward = fwd
        elif isinstance(m, nn.BatchNorm2d) and m not in no_round_bn:
            # a BN followed by ReLU is rounded after the ReLU by the HIP kernel; rounding commutes
            # with ReLU, so rounding the BN output here gives identical values
            m.register_forward_hook(lambda mod, inp, out: RoundBoth.apply(out))
        elif isinstance(m, (nn.MaxPool2d, tr.RefResBottleneck, tr.RefResBlock)):
            m.register_forward_hook(lambda mod, inp, out: RoundBoth.apply(out))
    feats = model.features

    def features_fwd(x, feats=feats):
        return nn.Sequential.forward(feats, RoundFwd.apply(x))     # NCHW fp32 -> bf16 hand-off
    model.features.forward = features_fwd
    return model
