"""Drop-in modules returned by the layer factory (convnets_amd/basemodel.py).

They SUBCLASS the stock torch.nn modules the reference's factory returns
(src/basemodel.py:966-1067), so everything the reference does with a layer keeps working:
`.kernel_size/.stride/.padding/.dilation/.out_channels` for the shape tracker (:725-781),
`isinstance(m, nn.Conv2d / nn.BatchNorm2d)` for init_params (:85-108), parameter/buffer names and
shapes for state_dict interchange (:883,940), `.train()/.eval()`, `.to(device)`, deepcopy.
Only `forward` is replaced: it dispatches to the HIP kernels through convnets_amd.ops.
"""
import weakref

import torch
import torch.nn as nn

from . import ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2d(nn.Conv2d):
    """nn.Conv2d with the MI355X implicit-GEMM forward/dgrad/wgrad (groups=1) or the depthwise
    kernels (groups == channels); dilation=1, zeros padding -- anything else is refused loudly."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._depthwise = self.groups > 1 and self.groups == self.in_channels == self.out_channels
        gw = self.in_channels // self.groups
        # grouped conv: block-diagonal per 64-channel chunk on the implicit-GEMM kernels when the shape allows it
        # (ResNeXt), else the dense conv of the block-diagonal weight (ShuffleNet's grouped 1x1 convs)
        self._chunked = self.groups > 1 and not self._depthwise and self.in_channels == self.out_channels \
            and self.in_channels % 64 == 0 and 64 % gw == 0
        self._dense_grouped = self.groups > 1 and not self._depthwise and not self._chunked
        dl = _pair(self.dilation)
        self._dil = dl[0]
        vec_ok = self.in_channels % 8 == 0 and self.out_channels % 8 == 0 and self.in_channels >= 32 \
            and self.out_channels >= 16 and not self._depthwise and not self._dense_grouped
        if dl[0] != dl[1] or (dl[0] != 1 and not vec_ok) or self.padding_mode != 'zeros' \
                or isinstance(self.padding, str):
            raise NotImplementedError('convnets_amd.Conv2d: zeros padding; dilation (equal in h and w) only on the '
                                      'implicit-GEMM shapes (channel counts that are multiples of 8, not depthwise)')
        # bf16 activations whose channel count is a multiple of 4 but not of 8 (ShuffleNet g1..g4: 36/60/68-wide
        # bottlenecks) have rows that are not whole 16-byte vectors, which the bf16 MFMA kernels need; the fp32
        # kernels take multiples of 4, so such a layer runs in fp32 between two casts instead of on the direct
        # kernels (50x slower): ShuffleNet-g4 106.9 -> see DESIGN.md
        ci, co = self.in_channels, self.out_channels
        self._f32_detour = not self._depthwise and not self._chunked and (ci % 8 != 0 or co % 8 != 0) \
            and ci % 4 == 0 and co % 4 == 0 and ci >= 16 and co >= 16 and dl[0] == 1
        self._packed = ops.PackedWeights()

    def __deepcopy__(self, memo):
        # the packed-weight cache is derived state; rebuild it in the copy
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = ops.PackedWeights() if k == '_packed' else copy.deepcopy(v, memo)
        return new

    def forward(self, x, want_bn_partial=False, link=None, fold=None):
        if self._depthwise:                      # conv2d_depthwise (basemodel.py:981-997)
            y = ops.DepthwiseConv2dFn.apply(x, self.weight, self.bias, _pair(self.stride), _pair(self.padding))
            return (y, torch.empty(0, device=x.device)) if want_bn_partial else y
        if self._f32_detour and x.dtype == torch.bfloat16 and link is None:
            w = ops.GroupedDenseWeightFn.apply(self.weight, self.groups) if self._dense_grouped else self.weight
            y = ops.Conv2dFn.apply(x.float(), w, self.bias, _pair(self.stride), _pair(self.padding), self._packed, False,
                                   None, None, 1 if self._dense_grouped else self.groups, self._dil)
            y = y.to(torch.bfloat16)
            return (y, torch.empty(0, device=x.device)) if want_bn_partial else y
        # x._cnk_bn: set when x is the output of a training-mode BatchNorm2d of this package (ops.BnRecord)
        if self._dense_grouped:
            wd = ops.GroupedDenseWeightFn.apply(self.weight, self.groups)
            return ops.Conv2dFn.apply(x, wd, self.bias, _pair(self.stride), _pair(self.padding),
                                      self._packed, want_bn_partial, link, getattr(x, '_cnk_bn', None), 1)
        return ops.Conv2dFn.apply(x, self.weight, self.bias, _pair(self.stride), _pair(self.padding),
                                  self._packed, want_bn_partial, link, getattr(x, '_cnk_bn', None), self.groups,
                                  self._dil, fold)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d; `forward(x, residual=None, relu=False)` exposes the fused epilogue used by
    ConvBlock and the residual blocks.  Called plainly it is exactly BatchNorm2d."""

    # set by MultiClassBaseModel.init_optimizer(): a [count] cell of pending training forwards; the
    # model then bumps every layer's num_batches_tracked with ONE kernel per step, not one per layer
    _deferred_counter = None

    def forward(self, x, residual=None, relu=False, partial=None, link=None, pool=None, fold=None, defer=False):
        """pool: the MaxPool2d module that consumes this layer's output (ConvBlock passes it for the stems, training
        mode only): BN, ReLU and the pooling then run as one autograd node and the POOLED tensor is returned."""
        if not (self.affine and self.track_running_stats):
            raise NotImplementedError('convnets_amd.BatchNorm2d: affine=True, '
                                      'track_running_stats=True only (the factory default)')
        if self.training:
            if self.momentum is None:
                raise NotImplementedError('convnets_amd.BatchNorm2d: cumulative momentum (None)')
            if pool is not None:
                y = ops.BatchNormActMaxPoolFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var,
                                                    float(self.eps), float(self.momentum), bool(relu), partial,
                                                    _pair(pool.kernel_size), _pair(pool.stride), _pair(pool.padding))
                y._cnk_pooled_by = pool
            else:
                rec = ops.BnRecord() if (torch.is_grad_enabled() and x.requires_grad and not defer) else None
                # residual operand whose own BN apply pass was deferred into this call (see `defer` below)
                res_affine = getattr(residual, '_cnk_defer', None) if residual is not None else None
                holder = [] if defer else None
                y = ops.BatchNormActFn.apply(x, self.weight, self.bias, residual, self.running_mean,
                                             self.running_var, float(self.eps), float(self.momentum),
                                             bool(relu), partial, link, rec, fold, holder, res_affine)
                if rec is not None and rec.x is not None:
                    y._cnk_bn = rec
                if defer:
                    # statistics only: y aliases x; whoever adds it as a residual normalises it on the fly
                    y._cnk_defer = holder[0]
            self._after_training_forward()
            return y
        if residual is not None and getattr(residual, '_cnk_defer', None) is not None:
            #  the residual is a projection shortcut's UN-normalised conv output whose BN apply pass was left to
            # this layer's training-mode pass; an eval-mode layer has no such pass -- adding it as it is would be silently wrong
            raise RuntimeError('convnets_amd.BatchNorm2d: the residual operand carries a deferred BN apply pass (a projection '
                               'shortcut in training mode) but this layer is in eval mode; put both in the same mode, or call '
                               'the shortcut block without defer_apply')
        return ops.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var,
                                   float(self.eps), residual, relu)


def _bn_after_training_forward(self):
    """Book-keeping of one training-mode forward (also called by ConvBlock when conv + BN ran as one node)."""
    self._stats_epoch = getattr(self, '_stats_epoch', 0) + 1     # running statistics changed
    if self._deferred_counter is not None:
        self._deferred_counter[0] += 1        # flushed for all BN layers at once by the model
    else:
        self.num_batches_tracked += 1


BatchNorm2d._after_training_forward = _bn_after_training_forward


class ReLU(nn.ReLU):
    """nn.ReLU(inplace=True) of the factory (basemodel.py:1012).  The result is returned in a new
    tensor: the reference's in-place flag is a memory optimisation with no observable effect on
    the values a Sequential passes on."""

    def forward(self, x):
        return ops.ReLUFn.apply(x)


class MaxPool2d(nn.MaxPool2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if _pair(self.dilation) != (1, 1) or self.ceil_mode or self.return_indices:
            raise NotImplementedError('convnets_amd.MaxPool2d: dilation=1, ceil_mode=False only')

    def forward(self, x):
        if getattr(x, '_cnk_pooled_by', None) is self:      # the ConvBlock in front already pooled (fused stem node)
            return x
        k = _pair(self.kernel_size)
        s = _pair(self.stride if self.stride is not None else self.kernel_size)
        return ops.MaxPool2dFn.apply(x, k, s, _pair(self.padding))


class AvgPool2d(nn.AvgPool2d):
    """nn.AvgPool2d of the factory (basemodel.py:1026-1033): count_include_pad=True, floor mode."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.ceil_mode or not self.count_include_pad or self.divisor_override is not None:
            raise NotImplementedError('convnets_amd.AvgPool2d: ceil_mode=False, count_include_pad=True only')

    def forward(self, x):
        k = _pair(self.kernel_size)
        s = _pair(self.stride if self.stride is not None else self.kernel_size)
        return ops.AvgPool2dFn.apply(x, k, s, _pair(self.padding))


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def __init__(self, output_size):
        super().__init__(output_size)
        if _pair(self.output_size) != (1, 1):
            raise NotImplementedError('convnets_amd.AdaptiveAvgPool2d: output_size=1 (global pool) only')

    def forward(self, x):
        return ops.GlobalAvgPoolFn.apply(x)


class Linear(nn.Linear):
    """nn.Linear of the classifier heads (resnet.py:106, vggnet.py:83-92) on the HIP fp32 GEMM."""

    def forward(self, x):
        return ops.LinearFn.apply(x, self.weight, self.bias)


class Dropout(nn.Dropout):
    """nn.Dropout(p) (resnet.py:105): identity in eval mode or at p == 0, HIP kernel otherwise."""

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if self.p >= 1.0:
            return x * 0.0
        return ops.DropoutFn.apply(x, self.p)       # fp32 (the head) or bf16 (feature maps), in memory order


class CrossEntropyLoss(nn.CrossEntropyLoss):
    """nn.CrossEntropyLoss(reduction='sum') as the reference builds it (basemodel.py:46).  Calling it
    returns the loss; `last_pred` then holds the per-sample argmax of the same logits (int32, device)
    for score() (basemodel.py:119-121)."""

    def __init__(self, reduction='sum'):
        super().__init__(reduction=reduction)
        if reduction != 'sum':
            raise NotImplementedError("convnets_amd.CrossEntropyLoss: reduction='sum' only (basemodel.py:46)")
        self.last_pred = None
        self.last_logits = None

    def forward(self, outputs, targets):
        loss, pred = ops.SoftmaxCrossEntropySumFn.apply(outputs, targets)
        # weak: a strong reference to the logits would keep the whole autograd graph of the step (and the
        # AccumulateGrad nodes of its leaves) alive until the next forward -- which breaks hipGraph capture
        self.last_pred, self.last_logits = pred, weakref.ref(outputs)
        return loss


class Sigmoid(nn.Sigmoid):
    def forward(self, x):
        return ops.SigmoidFn.apply(x)


class SEUnit(nn.Module):
    """Squeeze-and-excitation unit (src/senet.py:215-254, identical in src/se_resnet.py:216-255): global average
    pool -> Linear(C, C/r) -> ReLU -> Linear(C/r, C) -> Sigmoid -> per-channel scale.  Child names `squeeze` /
    `excitation.{0,2}` as in the reference.  The descriptor path runs in fp32 (GAP output), the scale on NHWC."""

    def __init__(self, network, reduction, **kwargs):
        super().__init__()
        num_features = network.in_channels
        self.squeeze = network.adapt_avgpool2d(set_output=False, output_size=1)
        self.excitation = nn.Sequential(Linear(num_features, num_features // reduction, bias=False), ReLU(inplace=True),
                                        Linear(num_features // reduction, num_features, bias=False), Sigmoid())

    def forward(self, x):
        n, c = x.shape[0], x.shape[1]
        vec = 8 if x.dtype == torch.bfloat16 else 4
        if x.is_cuda and c % vec == 0:
            # x is read by the squeeze and by the scale: their two gradients are written by ONE pass (ops.SKLink)
            link = ops.SKLink() if torch.is_grad_enabled() and x.requires_grad else None
            gate = self.excitation(ops.SKPoolFn.apply(link, x).view(n, c))
            return ops.ChannelScaleFn.apply(x, gate, link)
        gate = self.excitation(self.squeeze(x).view(n, c))
        return ops.ChannelScaleFn.apply(x, gate)


class SKConv(nn.Module):
    """Selective-kernel convolution (src/sknet.py:172-240, identical in src/sk_resnet.py:171-239): num_paths
    grouped 3x3 conv blocks with dilation 1+i, their sum squeezed by a global average pool into a descriptor
    (1x1 conv block), one 1x1 attention conv per path, softmax across the paths and the attention-weighted sum.
    Child names kernels / attentions / global_avgpool / descriptor / softmax as in the reference (same
    state_dict keys).  The fuse sum + pool and the softmax + weighted sum are one HIP pass each (csrc/sk.hip)."""

    def __init__(self, network, num_paths=2, groups=32, reduction=16, min_descriptor=32, **kwargs):
        super().__init__()
        self.num_paths = num_paths
        self.num_features = network.in_channels
        self.kernels = nn.ModuleList([])
        self.attentions = nn.ModuleList([])
        descriptor_size = max(network.in_channels // reduction, min_descriptor)
        for i in range(num_paths):
            self.kernels.append(Conv2dBlock(network, num_filters=network.in_channels, set_output=False, kernel_size=3,
                                            padding=1 + i, dilation=1 + i, groups=groups, **kwargs))
            self.attentions.append(nn.Sequential(Conv2d(descriptor_size, network.in_channels, kernel_size=1)))
        self.global_avgpool = network.adapt_avgpool2d(set_output=False, output_size=1)
        self.descriptor = Conv2dBlock(network, num_filters=descriptor_size, set_output=False, kernel_size=1)
        self.softmax = nn.Softmax(dim=1)              # kept for the module tree; the math runs in cnk_sk_select_fwd

    def forward(self, x):
        # x feeds every path: their data gradients are summed inside the last path's dgrad epilogue (ops.GradLink)
        # instead of by autograd -- which also lets that epilogue carry the backward reduction of the BN that made x
        glink = ops.GradLink(self.num_paths) if (self.num_paths > 1 and self.training and torch.is_grad_enabled()
                                                 and x.requires_grad) else None
        maps = [path(x, in_link=glink) for path in self.kernels]
        link = ops.SKLink() if torch.is_grad_enabled() and maps[0].requires_grad else None
        features = self.descriptor(ops.SKPoolFn.apply(link, *maps))
        logits = [attention(features) for attention in self.attentions]
        return ops.SKSelectFn.apply(link, self.num_paths, *maps, *logits)


import os as _os
DEFER_SHORTCUT_BN = _os.environ.get('CNK_DEFER_SHORTCUT_BN', '1') != '0'      # A/B switch
FUSED_BLOCK_NODE = _os.environ.get('CNK_FUSED_BLOCK_NODE', '1') != '0'        # A/B switch: conv + BN of a ConvBlock as one autograd node


class ConvBlock(nn.Sequential):
    """conv -> [BN] -> [ReLU] as conv2d_block builds it (basemodel.py:999-1015): an nn.Sequential
    with children '0','1','2' (same state_dict keys), whose forward runs BN + ReLU (+ the residual
    add of the enclosing block) as ONE fused kernel pass instead of three."""

    def forward(self, x, residual=None, relu_after_add=False, in_link=None, res_link=None, defer_apply=False):
        """in_link: GradLink shared with the other consumer of x (fused gradient sum at a residual
        block's input); res_link: the same link seen from the residual operand's side.
        defer_apply: conv -> BN without activation whose ONLY consumer is the residual add of another ConvBlock (a
        projection shortcut): training mode computes the statistics and returns the conv output itself, tagged with the
        layer's scale | shift; the consuming block's BN pass normalises it on the fly (bit-identical, one pass less)."""
        mods = list(self)
        conv = mods[0]
        bn = mods[1] if len(mods) > 1 and isinstance(mods[1], BatchNorm2d) else None
        act = isinstance(mods[-1], ReLU)
        if residual is not None and getattr(residual, '_cnk_defer', None) is not None and not (bn is not None and bn.training):
            raise RuntimeError('ConvBlock: a residual operand with a deferred BN apply pass needs a training-mode BatchNorm in this '
                               'block (the shortcut block asks its consumer first: ConvBlock.applies_deferred_residual)')
        if bn is not None:
            if bn.training:      # the conv epilogue also emits the BN statistics partials
                # (fold: the BN backward's apply pass runs inside this conv's backward where the shape allows, ops.FoldRecord)
                fold = ops.FoldRecord() if (ops.BN_FOLD and torch.is_grad_enabled()
                                            and isinstance(conv, Conv2d) and conv.kernel_size == (1, 1)) else None
                pool = self._linked_pool()
                defer = bool(defer_apply and not act and residual is None and torch.is_grad_enabled() and DEFER_SHORTCUT_BN)
                if FUSED_BLOCK_NODE and pool is None and torch.is_grad_enabled() and self._one_node_ok(conv, bn, x):
                    # conv -> BN (-> ReLU, + residual) as ONE autograd node (ops.ConvBnActFn): the same kernels in the same
                    # order, one Function.apply / autograd node / Module call chain less per block (host time of the eager step)
                    rec = ops.BnRecord() if (x.requires_grad or conv.weight.requires_grad) and not defer else None
                    res_affine = getattr(residual, '_cnk_defer', None) if residual is not None else None
                    holder = [] if defer else None
                    y = ops.ConvBnActFn.apply(
                        x, conv.weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var,
                        (_pair(conv.stride), _pair(conv.padding), conv._packed, in_link, getattr(x, '_cnk_bn', None), conv.groups,
                         conv._dil, fold),
                        (float(bn.eps), float(bn.momentum), bool(False if defer else (act or relu_after_add)),
                         None if defer else res_link, rec, holder, res_affine))
                    if rec is not None and rec.x is not None:
                        y._cnk_bn = rec
                    if defer:
                        y._cnk_defer = holder[0]
                    bn._after_training_forward()
                    return y
                y, partial = conv(x, want_bn_partial=True, link=in_link, fold=fold)
                if pool is not None and residual is None and torch.is_grad_enabled() and y.requires_grad:
                    # conv -> BN -> ReLU -> MaxPool2d (link_block_pools): one node, the pooled tensor comes back
                    return bn(y, relu=act, partial=partial, pool=pool)
                if defer:
                    return bn(y, partial=partial, defer=True, fold=fold if (fold is not None and fold.ok) else None)
                return bn(y, residual=residual, relu=act or relu_after_add, partial=partial,
                          link=res_link, fold=fold if (fold is not None and fold.ok) else None)
            if not torch.is_grad_enabled():
                # inference: BN folded into the conv, ReLU / residual add in its epilogue (one kernel)
                fold = self.__dict__.setdefault('_fold', ops.FoldedConv())
                stamp = (ops.weights_epoch(), getattr(bn, '_stats_epoch', 0), id(conv.weight), conv.weight._version,
                         bn.running_mean._version, bn.running_var._version, bn.weight._version, bn.bias._version)
                y = fold.run(x, conv.weight, conv.bias, bn, _pair(conv.stride), _pair(conv.padding), residual,
                             act or relu_after_add, stamp, conv.groups, conv._dil)
                if y is not None:
                    return y
            return bn(conv(x), residual=residual, relu=act or relu_after_add)
        y = conv(x, link=in_link)
        if residual is not None:
            if act:
                y = ops.ReLUFn.apply(y)
            return ops.AddReLUFn.apply(y, residual) if relu_after_add else y + residual
        return ops.ReLUFn.apply(y) if act else y


def _one_node_ok(self, conv, bn, x):
    """conv + BN of this block as ONE autograd node (ops.ConvBnActFn)?  The plain implicit-GEMM path only -- and only while
    nobody has hooked the two child modules, whose __call__ the fused node bypasses."""
    return (isinstance(conv, Conv2d) and conv.bias is None and not conv._depthwise and not conv._dense_grouped
            and not (conv._f32_detour and x.dtype == torch.bfloat16) and bn.affine and bn.track_running_stats
            and bn.momentum is not None and not conv._forward_hooks and not conv._forward_pre_hooks and not conv._backward_hooks
            and not bn._forward_hooks and not bn._forward_pre_hooks and not bn._backward_hooks)


ConvBlock._one_node_ok = _one_node_ok


def _applies_deferred_residual(self):
    """True when this block's forward will run a training-mode BN apply pass that can normalise a residual operand on the fly
    (cnk_bn_fwd_train_res_affine): what a projection shortcut asks its consumer before it defers its own apply pass -- with
    BN frozen per submodule the two layers may be in different modes."""
    mods = list(self)
    bn = mods[1] if len(mods) > 1 and isinstance(mods[1], BatchNorm2d) else None
    return bn is not None and bn.training and torch.is_grad_enabled() and DEFER_SHORTCUT_BN


ConvBlock.applies_deferred_residual = _applies_deferred_residual


def _conv_block_linked_pool(self):
    """The MaxPool2d that link_block_pools found behind this block -- re-validated at every call: the marker holds
    weak references to the parent Sequential and the pool, and is dropped as soon as the pool is no longer the
    module that follows this block there (child replaced or removed, block deep-copied or reused elsewhere)."""
    mark = self.__dict__.get('_next_pool')
    if mark is None:
        return None
    parent, pool = mark[0](), mark[1]()
    if parent is not None and pool is not None:
        kids = list(parent.children())
        for a, b in zip(kids, kids[1:]):
            if a is self:
                if b is pool:
                    return pool
                break
    del self.__dict__['_next_pool']
    return None


ConvBlock._linked_pool = _conv_block_linked_pool


def link_block_pools(model):
    """Marks every ConvBlock (with a BatchNorm) that is immediately followed by a MaxPool2d in an nn.Sequential --
    the stems of the ResNet-style families, VGG's stage ends, LeNet / AlexNet -- so that its training forward
    normalises inside the pooling pass (ops.BatchNormActMaxPoolFn).  The modules and their names stay as they
    are; the pool module sees a tensor its predecessor already pooled and passes it through."""
    import os
    if os.environ.get('CNK_FUSE_POOL', '1') == '0':          # A/B switch
        return
    for seq in model.modules():
        if isinstance(seq, nn.Sequential) and not isinstance(seq, ConvBlock):
            kids = list(seq.children())
            for a, b in zip(kids, kids[1:]):
                if isinstance(a, ConvBlock) and isinstance(b, MaxPool2d) and any(isinstance(m, BatchNorm2d) for m in a):
                    a.__dict__['_next_pool'] = (weakref.ref(seq), weakref.ref(b))


class Conv2dBlock(ConvBlock):
    """The reference's named-children variant (basemodel.py:1045-1067): conv / bn / relu."""

    def __init__(self, network, in_channels=None, num_filters=None, set_output=True, activation=True,
                 **kwargs):
        super().__init__()
        self.add_module('conv', network.conv2d(in_channels, num_filters, set_output, **kwargs))
        if network.setting.batch_norm:
            self.add_module('bn', BatchNorm2d(num_features=num_filters))
        if activation:
            self.add_module('relu', ReLU(inplace=True))


class Conv2dBlockDW(nn.Sequential):
    """MobileNet-v1's depthwise-separable block (src/mobilenet_v1.py:92-125), same child names
    (conv_dw, bn_dw, relu_dw, conv_pw, bn_pw, relu_pw => same state_dict keys): depthwise 3x3 ->
    [BN] -> [ReLU] -> pointwise 1x1 -> [BN] -> [ReLU], each BN+ReLU pair one fused kernel pass and the
    pointwise conv's dgrad epilogue carrying the depthwise BN's backward reduction."""

    def __init__(self, network, in_channels=None, num_filters=None, set_output=True, activation=True,
                 **kwargs):
        super().__init__()
        self.add_module('conv_dw', network.conv2d_depthwise(in_channels, in_channels, set_output, **kwargs))
        if network.setting.batch_norm:
            if in_channels is None:
                in_channels = network.in_channels
            self.add_module('bn_dw', BatchNorm2d(num_features=in_channels))
        if activation:
            self.add_module('relu_dw', ReLU(inplace=True))
        self.add_module('conv_pw', network.conv2d(in_channels, num_filters, set_output, kernel_size=1))
        if network.setting.batch_norm:
            self.add_module('bn_pw', BatchNorm2d(num_features=num_filters))
        if activation:
            self.add_module('relu_pw', ReLU(inplace=True))

    def forward(self, x):
        act = hasattr(self, 'relu_dw')
        y = self.conv_dw(x)
        if hasattr(self, 'bn_dw'):
            y = self.bn_dw(y, relu=act)
        elif act:
            y = ops.ReLUFn.apply(y)
        if hasattr(self, 'bn_pw'):
            if self.bn_pw.training:
                y, partial = self.conv_pw(y, want_bn_partial=True)
                return self.bn_pw(y, relu=act, partial=partial)
            return self.bn_pw(self.conv_pw(y), relu=act)
        y = self.conv_pw(y)
        return ops.ReLUFn.apply(y) if act else y
