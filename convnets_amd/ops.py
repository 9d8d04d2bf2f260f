"""torch.autograd.Function wrappers over the C ABI (include/cnk.h).

Tensors are logical NCHW in torch.channels_last memory format (= physical NHWC), so the
reference's semantics -- flatten order, state_dict shapes [Cout,Cin,kh,kw] -- are untouched
(reference call sites: src/basemodel.py:974,1008,1012,1021,1039; src/resnet.py:151-154,191-194).
PyTorch is used for device memory, streams and autograd bookkeeping only; every kernel that
touches an activation is in libcnk.so.  Nothing here falls back to ATen/MIOpen.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import CNK_BF16, CNK_F32, ConvDesc, PASS_DGRAD, PASS_FWD, PASS_WGRAD, check

CL = torch.channels_last

# bumped whenever parameters change behind autograd's back (flat-arena optimizer step,
# load_state_dict): conv modules re-pack their weight images when it moves
_weights_epoch = [0]


def bump_weights_epoch():
    _weights_epoch[0] += 1


def weights_epoch():
    return _weights_epoch[0]


# Called with a parameter whose .grad was written in place by a kernel (no autograd accumulation,
# hence no post-accumulate hook): the data-parallel reducer subscribes here.
_grad_ready_listeners = []


def _grad_written(param):
    for cb in _grad_ready_listeners:
        cb(param)


def _arena_grad(param):
    """The parameter's .grad if a kernel may accumulate into it directly (fp32, dense, same shape)."""
    if not param.is_leaf:              # e.g. the dense image of a grouped weight: autograd carries its gradient
        return None
    g = param.grad
    if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == param.shape \
            and g.device == param.device:
        return g
    return None


def _dtype_code(t):
    if t.dtype == torch.float32:
        return CNK_F32
    if t.dtype == torch.bfloat16:
        return CNK_BF16
    raise TypeError('convnets_amd: unsupported activation dtype %s (fp32 or bf16)' % t.dtype)


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('convnets_amd ops run on the MI355X HIP path only (got a %s tensor); '
                           'there is no CPU fallback' % t.device)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    # raw handle of the current stream of the current device (torch.cuda.current_stream() costs ~1 us of Python
    # per call, a tenth of the host time of a ResNet-50 step); a plain int: ctypes converts it for a c_void_p parameter
    # (0 = the null stream) without a wrapper object per call (round 5: ~2,600 pointer arguments per ResNet-152 step)
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _nhwc(t):
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


_workspaces = {}


def workspace(nbytes, device, stream=None, slot=0):
    """One grow-only scratch buffer per (device, stream); kernels are stream-ordered on the caller's
    stream, so successive ops on one stream can share it (the library itself never allocates).
    stream: the torch stream the kernels will run on when that is not the current one.  slot: a second buffer of the same
    stream (chained weight gradients alternate between two: the slabs of one launch are summed by the next)."""
    key = (device, stream.cuda_stream if stream is not None else
           _raw_stream(device.index) if (_raw_stream is not None and device.index is not None)
           else torch.cuda.current_stream(device).cuda_stream, slot)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        if stream is not None:
            # allocated from the CURRENT stream's pool but used on `stream`: when a bigger buffer replaces this one, the
            # allocator must not hand its memory on before that stream is done with it
            ws.record_stream(stream)
        _workspaces[key] = ws
    return ws


# ---- second HIP stream for the weight gradients -------------------------------------------------------
# In the backward of a conv, dgrad (needed by the next backward op) and wgrad (needed only by the
# optimizer / the gradient all-reduce) both depend on dy alone.  Both kernels are latency-bound and end
# in a partially filled last round of workgroups; issued on two streams they share the chip, each
# filling the other's idle slots.  The wgrad stream waits on an event recorded after dy is ready; whoever
# consumes weight gradients (FlatAdam.step, clipping, GradReducer) first calls join_side_streams().
WGRAD_STREAM = os.environ.get('CNK_WGRAD_STREAM', '1') != '0'
# Share of the split-K depth the weight gradients get while they run on the side stream (cnk_conv2d_wgrad: a workspace smaller
# than cnk_conv2d_workspace_bytes bounds the number of pixel-range splits, include/cnk.h).  The gradients only have to be done
# when the backward pass ends, and every CU holding one of their 112-KiB-LDS workgroups admits one implicit-GEMM workgroup of
# the compute stream instead of four: ResNet-50 / 152 step 1.7-2.3 % faster at 0.5, VGG-16 10 % slower (its weight gradients
# are the critical path) -- a property of the MODEL, so it is measured: MultiClassBaseModel.calibrate_wgrad_share() (the
# smallest candidate within 1 % of the fastest: eager probe steps of a host-bound model cannot tell the candidates apart).
WGRAD_SHARE = [float(os.environ.get('CNK_WGRAD_SHARE', '1.0'))]
_side_streams = {}
# tools/probe_graph_schedule.py: [uint64 device buffer, list of labels] -- device-clock stamps around the weight-gradient launches,
# readable after a graph replay (None: off, the product setting)
STAMP = [None]


def _stamp(label):
    buf, log = STAMP[0]
    if len(log) < buf.numel():
        _lib.load().cnk_debug_stamp(ctypes.c_void_p(buf.data_ptr() + 8 * len(log)), ctypes.c_void_p(_stream()))      # (a test hook without argtypes)
        log.append(label)



def wgrad_stream(device):
    """The device's weight-gradient stream: a HIP stream of our own (cnk_stream_create), wrapped for torch.  NOT
    torch.cuda.Stream(): those objects are dealt from a pool of 32 streams per device, round-robin, so after 32 more of them --
    one per GraphedTrainStep, per reducer, per test -- a "new" compute stream is this very stream: the step then runs on one
    stream, and the compute-stream scratch (keyed by stream handle) aliases the slabs a chained weight gradient has left
    for the next launch (found as a wrong hipGraph replay that only the full test suite could produce)."""
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = own_stream(device)
    return st


_comm_streams = {}


def comm_stream(device):
    """The device's stream for gradient collectives (dp.GradReducer), for the same reason a stream of our own."""
    st = _comm_streams.get(device)
    if st is None:
        st = _comm_streams[device] = own_stream(device)
    return st


def own_stream(device):
    """A HIP stream no other stream object of the process shares (see wgrad_stream); lives as long as the process."""
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(_lib.load().cnk_stream_create(ctypes.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=device)


def side_streams(device):
    st = _side_streams.get(device)
    return [st] if st is not None else []


_join_queued = [False]
# Chained slab sums (csrc/wgrad_chain.h): on the weight-gradient stream the fp32 slabs of conv k are summed by the workgroups of
# conv k + 1's kernel instead of by a launch of their own (52 latency-bound launches per ResNet-50 step: 0.4-0.55 ms of it);
# the last sum of a backward pass is flushed before the streams join.  device -> [pending descriptor, parameters to report, slot,
# tensors the descriptor points into].  The last entry keeps the slabs and the gradient buffer alive: a backward pass that
# raised half-way leaves its last sum pending, and the next pass (possibly of another model, after this one was freed) performs
# it -- into memory that must still belong to the dead step, not to whoever got it from the allocator since.
WGRAD_CHAIN = os.environ.get('CNK_WGRAD_CHAIN', '1') != '0'
_wgrad_chain = {}
_grad_sum_pending = set()              # id(parameter): its gradient kernel has run but the slab sum has not been issued yet


def grad_sum_pending(param):
    """True while `param.grad` is still waiting for a chained slab sum: autograd's own post-accumulate hook fires when the
    conv's backward returns, which is too early for anything that reads the gradient (dp.GradReducer); _grad_written follows."""
    return id(param) in _grad_sum_pending


def _flush_wgrad_chain(device):
    """Sums the slabs the last chained weight gradient left behind (on the weight-gradient stream) and reports its gradient."""
    st = _wgrad_chain.get(device)
    if st is None or st[0] is None:
        return
    pending, params = st[0], st[1]
    st[0], st[1] = None, ()
    check(_lib.load().cnk_wgrad_reduce_pending(ctypes.byref(pending), wgrad_stream(device).cuda_stream))
    st[3] = None
    for prm in params:
        _grad_sum_pending.discard(id(prm))
        _grad_written(prm)


_dy_event = {}                                # device -> the event eager steps fork the weight-gradient stream at
def _join_at_end_of_backward():
    """Once per backward pass: when the autograd engine has run the whole graph, the compute stream
    waits for the weight-gradient stream, so `.grad` is safe to read on the compute stream as usual."""
    if _join_queued[0]:
        return
    _join_queued[0] = True

    def cb():
        _join_queued[0] = False
        join_side_streams()
    torch.autograd.Variable._execution_engine.queue_callback(cb)


def _is_capturing(stream):
    with torch.cuda.stream(stream):
        return torch.cuda.is_current_stream_capturing()


def join_side_streams(device=None):
    """Makes the current stream wait for everything issued on the weight-gradient stream(s)."""
    _join_queued[0] = False                   # (whoever joins after a backward pass that raised: the flag must not stay set)
    for dev, st in _side_streams.items():
        if device is None or dev == device:
            _flush_wgrad_chain(dev)           # (the last chained slab sum has no launch behind it to ride in)
            cur = torch.cuda.current_stream(dev)
            if CAPTURE[0] is not None and not _is_capturing(st):
                continue              # hipGraph capture: nothing of this capture has been forked to that stream
            cur.wait_stream(st)


# Per-shape facts the library derives from a descriptor (output size, partial-slice counts, workspace bytes) are pure
# functions of the descriptor and of the library's switch table: they are asked for ONCE per (conv, input shape) and kept
# (round 3: six ctypes round trips and a struct construction less per conv call on the host; a ResNet-152 eager step issues
# ~1,900 launches from Python).  _lib.reload_switches() (tests, A/B tools) moves the epoch and drops them.
class ConvPlan:
    __slots__ = ('desc', 'Ho', 'Wo', 'fwd_slices', 'dgrad_slices', 'wgrad_ws', 'wgrad_min_ws', 'wgrad_min_ws_bias', 'accepts_mask',
                 'bnfold', 'fold_bytes', 'wgrad_chain', 'fold1', 'fold1_ws', 'fold1_slices')


_conv_plans = {}
_bn_sizes = {}


def conv_plan(x_shape, weight_shape, stride, padding, dtype_code, groups, dilation):
    key = (tuple(x_shape), tuple(weight_shape), tuple(stride), tuple(padding), dtype_code, groups, dilation, _lib.switch_epoch())
    pl = _conv_plans.get(key)
    if pl is None:
        lib = _lib.load()
        pl = ConvPlan()
        pl.desc = d = make_desc(x_shape, weight_shape, stride, padding, dtype_code, groups, dilation)
        Ho, Wo = ctypes.c_int32(), ctypes.c_int32()
        lib.cnk_conv2d_out_hw(ctypes.byref(d), ctypes.byref(Ho), ctypes.byref(Wo))
        pl.Ho, pl.Wo = Ho.value, Wo.value
        pl.fwd_slices = lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(d))
        pl.dgrad_slices = lib.cnk_conv2d_dgrad_bn_slices(ctypes.byref(d))
        pl.wgrad_ws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(d), PASS_WGRAD)
        pl.wgrad_min_ws = lib.cnk_conv2d_wgrad_min_workspace_bytes(ctypes.byref(d), 0)
        pl.wgrad_min_ws_bias = lib.cnk_conv2d_wgrad_min_workspace_bytes(ctypes.byref(d), 1)
        pl.accepts_mask = bool(lib.cnk_conv2d_dgrad_accepts_mask(ctypes.byref(d)))
        pl.bnfold = bool(lib.cnk_conv2d_bnfold_supported(ctypes.byref(d)))
        pl.fold_bytes = lib.cnk_conv2d_bnfold_image_bytes(ctypes.byref(d)) if pl.bnfold else 0
        pl.wgrad_chain = bool(lib.cnk_conv2d_wgrad_chain_supported(ctypes.byref(d)))
        pl.fold1 = pl.bnfold and bool(lib.cnk_conv2d_bwd_bnfold_supported(ctypes.byref(d)))
        pl.fold1_ws = lib.cnk_conv2d_bwd_bnfold_workspace_bytes(ctypes.byref(d)) if pl.fold1 else 0
        pl.fold1_slices = lib.cnk_conv2d_bwd_bnfold_bn_slices(ctypes.byref(d)) if pl.fold1 else 0
        if len(_conv_plans) > 4096:
            _conv_plans.clear()
        _conv_plans[key] = pl
    return pl


def bn_sizes(M, C, dtype_code):
    """(workspace bytes, mask bytes) of a BN layer over [M][C]."""
    key = (M, C, dtype_code)
    v = _bn_sizes.get(key)
    if v is None:
        lib = _lib.load()
        v = _bn_sizes[key] = (lib.cnk_bn_workspace_bytes(M, C), lib.cnk_bn_mask_bytes(M, C, dtype_code))
    return v


def make_desc(x_shape, weight_shape, stride, padding, dtype_code, groups=1, dilation=1):
    N, Cin, H, W = x_shape
    Cout, Cin_w, kh, kw = weight_shape
    if Cin != Cin_w * groups:
        raise ValueError('conv2d: input has %d channels, weight expects %d' % (Cin, Cin_w * groups))
    return ConvDesc(N, H, W, Cin, Cout, kh, kw, stride[0], stride[1], padding[0], padding[1],
                    dtype_code, groups, dilation)


def masked_grad(dy, mask):
    """dy * relu_mask for a packed mask (fallback when a consumer cannot apply the mask itself)."""
    N, C, H, W = dy.shape
    vec = dy.numel() // mask.numel()
    bits = (mask.view(-1, 1) >> torch.arange(vec, device=mask.device, dtype=torch.uint8)) & 1
    m = bits.view(N, H, W, C).permute(0, 3, 1, 2)
    return (dy * m.to(dy.dtype)).contiguous(memory_format=CL)


class BnRecord:
    """Travels with a training-mode BatchNormActFn output y (as y._cnk_bn) to the conv that consumes
    it.  That conv's data gradient IS the gradient w.r.t. y, so its epilogue can run the first pass
    of this BN layer's backward (ReLU masking + the two per-channel sums): the conv's backward fills
    `fused = (dx, partial)`, and BatchNormActFn.backward uses it only if the gradient autograd then
    delivers is that very tensor (any other consumer of y makes autograd sum into a new tensor, and
    the regular two-pass backward runs: masking twice is idempotent, so that path stays exact)."""
    __slots__ = ('x', 'mean', 'invstd', 'gamma', 'beta', 'mask', 'relu', 'fused')

    def __init__(self):
        self.x = self.mean = self.invstd = self.gamma = self.beta = self.mask = self.fused = None
        self.relu = False


class FoldRecord:
    """Shared by the Conv2dFn and the BatchNormActFn of one conv -> BN pair (layers.ConvBlock.forward; the conv output is a
    local of that forward, so the BN layer is its only consumer).  For a dense 1x1 / stride-1 bf16 conv the BN backward's apply
    pass is folded into the conv's own backward (csrc/bnfold.hip): BatchNormActFn.backward then only finalises the partial sums
    (dgamma, dbeta, the three coefficient vectors), parks them here together with the gradient w.r.t. the BN OUTPUT, and hands
    that very tensor to autograd as the "gradient" of the conv output; Conv2dFn.backward recognises it (same storage) and runs
    cnk_conv2d_dgrad_bnfold / cnk_conv2d_wgrad_bnfold.  The [M][Cout] gradient w.r.t. the conv output is never written."""
    __slots__ = ('ok', 'g', 'y', 'mean', 'invstd', 'coef')

    def __init__(self):
        self.ok = False                   # set by Conv2dFn.forward when the conv shape can fold
        self.g = self.y = self.mean = self.invstd = self.coef = None


BN_FOLD = os.environ.get('CNK_BN_FOLD', '1') != '0'
# Round 5: where the library has it (64 -> 256 channels: the first ResNet stage, where the backward is HBM-bound on both streams),
# the folded data gradient and the folded weight gradient are ONE pass over g and the conv input on the compute stream
# (cnk_conv2d_bwd_bnfold, csrc/conv_fold3.hip) -- the weight-gradient stream loses its 925-MB launches.
FOLD_ONE_PASS = os.environ.get('CNK_FOLD_ONE_PASS', '1') != '0'
# The fold pays where the apply pass it removes (3 * M * Cout elements of HBM traffic) outweighs what it adds to the critical
# path (the finalise + image launches, ~20-30 us, and K + Cin/Cout in the data gradient) and to the weight-gradient stream (the
# BN input read beside the gradient): measured per ResNet-50 stage at batch 256 (tools/bench_fold.py) 366 -> 163 / 203 -> 110 /
# 119 -> 83 / 87 -> 91 us on the compute stream, i.e. not at 7x7; the rule below is M * Cout elements of the BN tensor.
BN_FOLD_MIN_ELEMS = float(os.environ.get('CNK_BN_FOLD_MIN', '4e7'))
# (Round 4 also had: the fold's second form for convs that narrow (the BN input as second K segment), the fold of strided
# projection shortcuts, and LazyConvOut (the conv output of a bottleneck's last block left out of the forward and replayed in
# the backward).  Each measured slower or neutral in the ResNet-50 step -- 19.80 -> 19.95, 19.23 vs 19.26, 19.65 -> 19.88 ms,
# DESIGN.md 7b / 9 -- and was removed in round 5: one folded path in Conv2dFn.backward.)
FOLD_COUNTS = {'folded': 0, 'one_pass': 0}               # folded conv backward passes (tests read this)
BN_BWD_FUSE = os.environ.get('CNK_BN_BWD_FUSE', '1') != '0'
# the same fusion for the stride-1 3x3 convs only (A/B switch, DESIGN.md: break-even per layer, see there)
BN_BWD_FUSE_3X3 = os.environ.get('CNK_BN_BWD_FUSE_3X3', '1') != '0'
FUSE_COUNTS = {'bn_bwd_fused': 0, 'bn_bwd_unfused': 0}      # which backward path ran (tests read this)


class GradLink:
    """Hands the gradient of one consumer of a tensor to another consumer's backward so the sum of
    the two is formed inside a kernel epilogue instead of by autograd's accumulation pass.
    Used at a residual block's input x, consumed by the block's first conv and by the shortcut
    (identity: the last BN's residual operand; projection: the dim_synch conv).  Whichever
    backward runs first parks its gradient here and reports None to autograd; the one that runs
    last folds the parked tensor in (conv dgrad: fused `addend`) and returns the total."""

    def __init__(self, count=2):
        self.count, self.arrived, self.parked, self.final = count, 0, None, False

    def contribute(self, make_grad):
        """make_grad(addend_or_None) -> gradient tensor.  Returns the tensor to report to autograd
        (None for all but the last contributor)."""
        self.arrived += 1
        self.final = self.arrived >= self.count       # make_grad may ask: is this the total gradient?
        if self.arrived < self.count:
            g = make_grad(None)
            if self.parked is None:
                self.parked = g
            else:
                a, b = self.parked, g
                a = masked_grad(*a) if isinstance(a, tuple) else a
                b = masked_grad(*b) if isinstance(b, tuple) else b
                self.parked = a + b
            return None
        parked, self.arrived, self.parked = self.parked, 0, None
        return make_grad(parked)


class WeightPacker:
    """Re-lays the weight images of EVERY registered conv with one kernel launch
    (cnk_pack_batch) per weights epoch, instead of two to five small launches per conv."""

    def __init__(self):
        self._refs = []            # weak references to the PackedWeights that asked to be batched
        self.table_key = None
        self.jobs_dev = self.map_dev = None
        self.nblocks = 0
        self.done_epoch = -1

    def register(self, pw):
        import weakref
        if not any(r() is pw for r in self._refs):
            self._refs.append(weakref.ref(pw))      # weak: a packer must not keep dead models alive
            self.table_key = None

    @property
    def entries(self):
        live = [r() for r in self._refs]
        if any(p is None for p in live):
            self._refs = [r for r in self._refs if r() is not None]
            self.table_key = None
        return [p for p in live if p is not None]

    def _build(self, entries):
        lib = _lib.load()
        jobs, blocks = [], []
        buf = (_lib.PackJob * 24)()
        for pw in entries:
            n = lib.cnk_conv2d_pack_jobs(ctypes.byref(pw.desc), _p(pw.weight), _p(pw.w_fwd), _p(pw.w_dgrad),
                                         buf, 24)
            if n < 0:
                raise RuntimeError('WeightPacker: conv images cannot be batched')
            for k in range(n):
                j = _lib.PackJob()
                ctypes.memmove(ctypes.byref(j), ctypes.byref(buf[k]), ctypes.sizeof(j))
                first = len(blocks)
                blocks += [(len(jobs), first)] * j.nblocks
                jobs.append(j)
        dev = entries[0].weight.device
        raw = b''.join(bytes(j) for j in jobs)
        self.jobs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.map_dev = torch.tensor(blocks, dtype=torch.int32).to(dev)
        self.nblocks = len(blocks)

    def sync_table(self):
        """Rebuilds the device-side job table NOW if the set of registered convs changed (a model was freed, a new one
        registered).  GraphedTrainStep calls it right before a capture: the rebuild uploads a table from the host,
        which is not allowed while a stream is capturing."""
        entries = self.entries
        if not entries:
            return
        key = tuple((pw.weight.data_ptr(), pw.w_fwd.data_ptr(),
                     0 if pw.w_dgrad is None else pw.w_dgrad.data_ptr()) for pw in entries)
        if key != self.table_key:
            self._build(entries)
            self.table_key = key

    def ensure(self):
        """Called by a registered PackedWeights whose images are stale: repack ALL of them."""
        ep = weights_epoch()
        if self.done_epoch == ep:
            return
        entries = self.entries
        if not entries:
            return
        key = tuple((pw.weight.data_ptr(), pw.w_fwd.data_ptr(),
                     0 if pw.w_dgrad is None else pw.w_dgrad.data_ptr()) for pw in entries)
        if key != self.table_key:
            self._build(entries)
            self.table_key = key
        with torch.no_grad():
            check(_lib.load().cnk_pack_batch(_p(self.jobs_dev), _p(self.map_dev), self.nblocks, _stream()))
        self.done_epoch = ep
        for pw in entries:
            pw.batched_epoch = ep


_packers = {}


def packer_for(device):
    p = _packers.get(device)
    if p is None:
        p = _packers[device] = WeightPacker()
    return p


class PackedWeights:
    """The two operand images of one conv weight (cnk_conv2d_pack_weights), re-made when the
    master weight changes.  After its first use a conv whose images can be batched registers with
    the device's WeightPacker, which from then on re-lays all registered convs in one launch."""

    def __init__(self):
        self.key = None
        self.w_fwd = None
        self.w_dgrad = None
        self.weight = self.desc = None
        self.batched = False
        self.batched_epoch = -1

    def get(self, weight, desc, need_dgrad):
        lib = _lib.load()
        if self.batched and self.weight is weight and self.key is not None \
                and self.key[0] == weight.data_ptr() and self.key[3] == desc.dtype and self.key[4] == need_dgrad \
                and self.key[1] == weight._version:
            if self.batched_epoch != weights_epoch():
                packer_for(weight.device).ensure()
            self.key = (weight.data_ptr(), weight._version, weights_epoch(), desc.dtype, need_dgrad, desc.groups, desc.dilation)
            return self.w_fwd, self.w_dgrad
        key = (weight.data_ptr(), weight._version, weights_epoch(), desc.dtype, need_dgrad, desc.groups, desc.dilation)
        if self.key != key:
            dev = weight.device
            nf = lib.cnk_conv2d_packed_bytes(ctypes.byref(desc), PASS_FWD)
            self.w_fwd = torch.empty(nf, dtype=torch.uint8, device=dev)
            self.w_dgrad = None
            if need_dgrad:
                nd = lib.cnk_conv2d_packed_bytes(ctypes.byref(desc), PASS_DGRAD)
                self.w_dgrad = torch.empty(nd, dtype=torch.uint8, device=dev)
            with torch.no_grad():
                check(lib.cnk_conv2d_pack_weights(ctypes.byref(desc), _p(weight), _p(self.w_fwd),
                                                  _p(self.w_dgrad), _stream()))
            self.key = key
            # a non-leaf weight (the dense image of a grouped weight, re-made every forward) is neither kept -- the
            # reference would keep its autograd graph, and the AccumulateGrad node of an earlier iteration, alive --
            # nor batched (its storage changes from step to step)
            self.weight, self.desc = (weight if weight.is_leaf else None), desc
            # batch from the next epoch on if every image of this conv is an implicit-GEMM image
            probe = (_lib.PackJob * 24)()
            self.batched = weight.is_leaf and weight.requires_grad and lib.cnk_conv2d_pack_jobs(
                ctypes.byref(desc), _p(weight), _p(self.w_fwd), _p(self.w_dgrad), probe, 24) >= 0
            if self.batched:
                self.batched_epoch = weights_epoch()
                packer_for(weight.device).register(self)
        return self.w_fwd, self.w_dgrad


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d forward / dgrad / wgrad (reference: basemodel.py:974)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, packed, want_bn_partial=False, link=None,
                bn_rec=None, groups=1, dilation=1, fold=None):
        """Returns y, or (y, bn_partial) when want_bn_partial: the per-tile BN statistics partials
        of y emitted by the conv epilogue (None when this conv shape cannot produce them).
        link: optional GradLink shared with the other consumer of x (residual blocks)."""
        _require_gpu(x)
        lib = _lib.load()
        ctx.set_materialize_grads(False)      # no zero tensor for the (non-differentiable) bn_partial
        x_needs_grad = x.requires_grad        # before any layout copy made here (forward runs under no_grad)
        x = _nhwc(x)
        w = weight if weight.is_contiguous() else weight.contiguous()
        plan = conv_plan(x.shape, w.shape, stride, padding, _dtype_code(x), groups, dilation)
        desc = plan.desc
        need_dgrad = x_needs_grad
        w_fwd, w_dgrad = packed.get(w, desc, need_dgrad)
        y = torch.empty((desc.N, desc.Cout, plan.Ho, plan.Wo), dtype=x.dtype, device=x.device,
                        memory_format=CL)
        partial = None
        if want_bn_partial:
            slices = plan.fwd_slices
            if slices > 0:
                partial = torch.empty((slices, desc.Cout, 3), dtype=torch.float32, device=x.device)
        check(lib.cnk_conv2d_fwd(ctypes.byref(desc), _p(x), _p(w_fwd), _p(bias), _p(y), _p(partial),
                                 None, 0, _stream()))
        ctx.save_for_backward(x, w)
        ctx.desc, ctx.w_dgrad, ctx.has_bias, ctx.plan = desc, w_dgrad, bias is not None, plan
        ctx.params = (weight, bias)
        ctx.link = link if x_needs_grad else None
        ctx.bn_rec = bn_rec if (x_needs_grad and BN_BWD_FUSE) else None
        # conv -> BN with the BN backward folded into this conv's backward (FoldRecord): dense 1x1 bf16, no bias
        ctx.fold = None
        if fold is not None and BN_FOLD and plan.bnfold and bias is None and w.dtype == torch.float32 \
                and x_needs_grad and w_dgrad is not None:
            elems = desc.N * plan.Ho * plan.Wo * desc.Cout
            if desc.Cout >= 2 * desc.Cin and elems >= BN_FOLD_MIN_ELEMS:
                fold.ok = True                        # K grows by Cin / Cout
                ctx.fold = fold
        if want_bn_partial:
            if partial is None:
                partial = torch.empty(0, device=x.device)         # placeholder: "not available"
            ctx.mark_non_differentiable(partial)
            return y, partial
        return y

    @staticmethod
    def backward(ctx, dy, _dpartial=None):
        lib = _lib.load()
        if dy is None:                        # y took no part in the loss
            return (None,) * 12
        x, w = ctx.saved_tensors
        desc = ctx.desc
        # folded BN backward: dy is the gradient w.r.t. the BN OUTPUT, parked by BatchNormActFn.backward
        fold, ctx.fold = ctx.fold, None
        if fold is not None:
            fold.ok = False
            if fold.g is None:
                fold = None                   # the BN backward in front of this node took the unfolded path: dy is dy_conv
            elif fold.g.data_ptr() != dy.data_ptr() or fold.g.shape != dy.shape:
                # BatchNormActFn.backward DID fold (it skipped its apply pass and returned the gradient w.r.t. the BN output
                # as this conv's "gradient"), but autograd delivers a different tensor -- a hook cloned or rescaled it, or a
                # second consumer's gradient was accumulated on top: the plain backward on it would be silently wrong
                # 
                fold.g = fold.y = fold.mean = fold.invstd = fold.coef = None
                raise RuntimeError('convnets_amd: the gradient of a conv output whose BN backward was folded into the conv '
                                   '(ops.FoldRecord) reached Conv2dFn.backward as a different tensor (a tensor hook or a second '
                                   'consumer of the conv output?); run with CNK_BN_FOLD=0')
        dy = _nhwc(dy)
        dx = dw = db = None
        w_fold = bias_fold = None
        if fold is not None:
            FOLD_COUNTS['folded'] += 1
            w_fold = torch.empty(ctx.plan.fold_bytes, dtype=torch.uint8, device=x.device)
            bias_fold = torch.empty(desc.Cin, dtype=torch.float32, device=x.device)
            check(lib.cnk_conv2d_bnfold_prepare(ctypes.byref(desc), _p(w), _p(ctx.w_dgrad), _p(fold.mean), _p(fold.invstd),
                                                _p(fold.coef), _p(w_fold), _p(bias_fold), _stream()))
        # dy is complete on the compute stream HERE: the weight gradient (issued below on the second
        # stream) waits for this point only, so it also overlaps this conv's own data gradient
        # (Round 5: every such event is a fork of the captured graph, and the compute-stream kernel behind a fork starts ~4.4 us
        # late -- 50-odd gaps per step in the kernel timeline.  Forking once per N convs, the weight gradients in between
        # parked, was measured: 19.09-19.30 ms at N = 1, 19.22-19.28 at 2, 19.39 at 3, 19.43 at 4, 19.5-19.7 at 8 -- the later
        # start of the side stream costs more than the gaps; not kept.)
        dy_ready = None
        # (bf16 path only: its kernels are latency/HBM-bound and gain from sharing the chip; the fp32
        # kernels run at 65-70 % of the fp32 MFMA peak and only contend -- AlexNet fp32 lost 4 %)
        if WGRAD_STREAM and x.is_cuda and x.dtype == torch.bfloat16:
            # (eager steps re-record ONE event per device: a stream's wait refers to the record it was issued behind; a
            # capture gets an event per fork)
            dy_ready = _dy_event.get(x.device) if CAPTURE[0] is None else None
            if dy_ready is None:
                dy_ready = torch.cuda.Event()
                if CAPTURE[0] is None:
                    _dy_event[x.device] = dy_ready
            dy_ready.record(torch.cuda.current_stream(x.device))
        wgrad_done = False
        if ctx.needs_input_grad[0]:
            def dgrad(addend):
                nonlocal wgrad_done
                amask = None
                if isinstance(addend, tuple):        # (block-output gradient, its packed ReLU mask)
                    addend, amask = addend
                    if not ctx.plan.accepts_mask:
                        addend, amask = masked_grad(addend, amask), None
                if addend is not None:
                    addend = _nhwc(addend)
                out = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=CL)
                # x is the output of a BN(+ReLU) layer and `out` the TOTAL gradient w.r.t. it: run the
                # first pass of that layer's backward in this epilogue (BnRecord)
                rec, fuse, partial = ctx.bn_rec, None, None
                fuse_ok = rec is not None and (ctx.link is None or ctx.link.final) and rec.x is not None \
                    and rec.x.shape == x.shape and rec.x.dtype == x.dtype \
                    and (BN_BWD_FUSE_3X3 or desc.kh != 3 or desc.stride_h != 1)
                # both folded gradients in one pass: a gradient addend OR the fused reduction (with a re-derived mask), arena weight gradient
                gw1 = None
                if fold is not None and FOLD_ONE_PASS and ctx.plan.fold1 and amask is None and not ctx.has_bias \
                        and ctx.needs_input_grad[1] and not (fuse_ok and (rec.mask is not None or addend is not None)) \
                        and ctx.params[0].is_contiguous():
                    gw1 = _arena_grad(ctx.params[0])
                if fuse_ok:
                    slices = ctx.plan.fold1_slices if gw1 is not None else ctx.plan.dgrad_slices
                    if slices > 0:
                        partial = torch.empty((slices, desc.Cin, 2), dtype=torch.float32, device=x.device)
                        fuse = _lib.BnBwdFuse(rec.x.data_ptr(), rec.mean.data_ptr(), rec.invstd.data_ptr(),
                                              rec.gamma.data_ptr(), rec.beta.data_ptr(),
                                              rec.mask.data_ptr() if rec.mask is not None else None,
                                              partial.data_ptr(), int(rec.relu))
                if gw1 is not None:
                    ws1 = workspace(ctx.plan.fold1_ws, x.device)
                    check(lib.cnk_conv2d_bwd_bnfold(ctypes.byref(desc), _p(dy), _p(x), _p(w_fold), _p(bias_fold), _p(out), _p(addend),
                                                    ctypes.byref(fuse) if fuse else None, _p(w), _p(fold.mean), _p(fold.invstd),
                                                    _p(fold.coef), _p(gw1), 1, _p(ws1), ws1.numel(), _stream()))
                    FOLD_COUNTS['one_pass'] += 1
                    wgrad_done = True
                    _grad_written(ctx.params[0])
                elif fold is not None:
                    check(lib.cnk_conv2d_dgrad_bnfold(ctypes.byref(desc), _p(dy), _p(x),
                                                      _p(w_fold), _p(bias_fold), _p(out), _p(addend), _p(amask),
                                                      ctypes.byref(fuse) if fuse else None, _stream()))
                else:
                    check(lib.cnk_conv2d_dgrad(ctypes.byref(desc), _p(dy), _p(ctx.w_dgrad), _p(out),
                                               _p(addend), _p(amask), ctypes.byref(fuse) if fuse else None,
                                               None, 0, _stream()))
                if fuse is not None:
                    rec.fused = (out, partial)
                return out
            dx = ctx.link.contribute(dgrad) if ctx.link is not None else dgrad(None)
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and not wgrad_done:
            weight, bias = ctx.params
            # accumulate straight into .grad when it is a dense fp32 buffer (the flat arena): same
            # result as autograd's `grad += dw`, without the extra kernel and tensor per parameter
            gw = _arena_grad(weight) if weight.is_contiguous() else None
            gb = _arena_grad(bias) if ctx.has_bias else None
            direct = gw is not None and (not ctx.has_bias or gb is not None)
            if direct:
                dw_buf, db_buf = gw, gb
            else:
                dw_buf = dw = torch.empty(w.shape, dtype=torch.float32, device=w.device)
                db_buf = db = torch.empty((desc.Cout,), dtype=torch.float32, device=w.device) \
                    if ctx.has_bias else None
            nws = ctx.plan.wgrad_ws
            min_ws = ctx.plan.wgrad_min_ws_bias if ctx.has_bias else ctx.plan.wgrad_min_ws

            fy, fmean, finv, fcoef = (fold.y, fold.mean, fold.invstd, fold.coef) if fold is not None else (None,) * 4

            def run_wgrad(share=1.0, stream=None):
                # stream: the weight-gradient stream, WITHOUT making it torch's current stream (`with torch.cuda.stream(..)`
                # costs ~10 us of host time per conv: 1.5 ms of a ResNet-152 step)
                st = _stream() if stream is None else stream.cuda_stream
                chain = WGRAD_CHAIN and stream is not None and direct and not ctx.has_bias and ctx.plan.wgrad_chain
                cst = _wgrad_chain.setdefault(x.device, [None, (), 0, None]) if stream is not None else None
                if chain:
                    cst[2] ^= 1               # the other workspace: the pending slabs sit in the one the last launch used
                elif cst is not None:
                    _flush_wgrad_chain(x.device)          # (an unchained launch on that stream would overwrite pending slabs)
                ws = workspace(nws, x.device, stream, cst[2] if chain else 0)
                budget = ws.numel()
                # a share of the split-K depth: never below what the library says it needs (the whole workspace for every
                # kernel that does not split; cnk_conv2d_wgrad_min_workspace_bytes)
                if share < 1.0:
                    budget = min(budget, max(min_ws, int(nws * share) // 256 * 256))
                if chain:
                    # this launch sums the slabs of the PREVIOUS weight gradient of the stream in its prologue and leaves its own
                    prev, prev_params = cst[0], cst[1]
                    out = _lib.WgradPending()
                    if fy is not None:
                        check(lib.cnk_conv2d_wgrad_bnfold_chain(ctypes.byref(desc), _p(dy), _p(fy), _p(x), _p(fmean), _p(finv),
                                                                _p(fcoef), _p(dw_buf), 1, _p(ws), budget,
                                                                ctypes.byref(prev) if prev is not None else None, ctypes.byref(out), st))
                    else:
                        check(lib.cnk_conv2d_wgrad_chain(ctypes.byref(desc), _p(dy), _p(x), _p(dw_buf), 1, _p(ws), budget,
                                                         ctypes.byref(prev) if prev is not None else None, ctypes.byref(out), st))
                    cst[0], cst[1], cst[3] = out, (weight,), (dw_buf, ws)
                    _grad_sum_pending.add(id(weight))
                    for prm in prev_params:
                        _grad_sum_pending.discard(id(prm))
                        _grad_written(prm)    # its sum has been issued (inside the launch above)
                    return
                if fy is not None:
                    check(lib.cnk_conv2d_wgrad_bnfold(ctypes.byref(desc), _p(dy), _p(fy), _p(x), _p(fmean),
                                                      _p(finv), _p(fcoef), _p(dw_buf), int(direct), _p(ws),
                                                      budget, st))
                else:
                    check(lib.cnk_conv2d_wgrad(ctypes.byref(desc), _p(dy), _p(x), _p(dw_buf), _p(db_buf),
                                               int(direct), _p(ws), budget, st))
                if direct:
                    _grad_written(weight)
                    if ctx.has_bias:
                        _grad_written(bias)
            if dy_ready is not None and direct:
                # in-place arena gradients only: a returned dw tensor would be consumed by autograd on
                # the main stream without knowing about the side stream
                side = wgrad_stream(x.device)
                # (ORDER MATTERS under capture: the data gradient above was launched on the compute stream BEFORE the first launch
                # on the side stream below.  The graph executor derives its hardware queues from the creation order of a fork
                # node's children; with the side-stream launch first, replays took 26-30 ms instead of 18.6 -- DESIGN.md 7c;
                # cnk_graph_anchor is the remedy measured for that pattern)
                side.wait_event(dy_ready)                                    # dy, x (and zero_grad) are ready
                if STAMP[0] is not None:                                     # (tools/probe_graph_schedule.py: the slow way)
                    _stamp('main')
                    with torch.cuda.stream(side):
                        _stamp('side_begin')
                        run_wgrad(WGRAD_SHARE[0])
                        _stamp('side_end')
                else:
                    run_wgrad(WGRAD_SHARE[0], side)
                dy.record_stream(side)
                x.record_stream(side)
                if fy is not None:
                    for t in (fy, fmean, finv, fcoef):
                        t.record_stream(side)
                _join_at_end_of_backward()
            else:
                run_wgrad()
        if fold is not None:
            fold.g = fold.y = fold.mean = fold.invstd = fold.coef = None
        return dx, dw, db, None, None, None, None, None, None, None, None, None


class BatchNormActFn(torch.autograd.Function):
    """nn.BatchNorm2d (training mode) fused with the following ReLU and, in the residual blocks,
    with `output += identity; F.relu(output)` (basemodel.py:1008-1012, resnet.py:191-194)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, relu,
                partial=None, link=None, rec=None, fold=None, defer=None, res_affine=None, x_needs_grad=None):
        """x_needs_grad: set by ConvBnActFn, whose conv half produced x inside the same node (a tensor made under the
        Function's no_grad carries no requires_grad flag).
        defer (a list, or None): statistics only -- the apply pass of this layer (no activation, no residual: a projection
        shortcut, resnet.py:90-91) is left to the ONE consumer of its output, the `output += identity` of the block's last
        BN; the returned tensor aliases x (un-normalised) and the layer's scale | shift [2][C] are appended to the list
        (layers.BatchNorm2d.forward attaches them to the output as `_cnk_defer`).  res_affine: such a deferred layer's
        scale | shift for the residual operand of THIS call (cnk_bn_fwd_train_res_affine): bit-identical to the two-pass
        sequence, without writing and re-reading the normalised shortcut tensor."""
        _require_gpu(x)
        lib = _lib.load()
        if x_needs_grad is None:
            x_needs_grad = x.requires_grad
        res_needs_grad = residual is not None and residual.requires_grad
        x = _nhwc(x)
        if residual is not None:
            residual = _nhwc(residual)
            if residual.shape != x.shape or residual.dtype != x.dtype:
                raise ValueError('batch_norm: residual must match the input shape and dtype')
        N, C, H, W = x.shape
        M = N * H * W
        if defer is not None and (relu or residual is not None):
            raise ValueError('batch_norm: a deferred apply pass takes neither an activation nor a residual operand')
        save_mean = torch.empty(C, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws_bytes, mask_bytes = bn_sizes(M, C, _dtype_code(x))
        ws = workspace(ws_bytes, x.device)
        if partial is not None and partial.numel() == 0:
            partial = None
        nsl = 0 if partial is None else partial.shape[0]
        # with a residual operand the ReLU mask cannot be re-derived from x alone: keep it as a
        # packed bit mask (1/16 of y's bytes) for the backward instead of re-reading y twice
        mask = None
        if relu and residual is not None and x_needs_grad:
            mask = torch.empty(mask_bytes, dtype=torch.uint8, device=x.device)
        if defer is not None:
            # statistics only; scale | shift are left at the start of the (shared) workspace: keep a copy for the consumer
            check(lib.cnk_bn_fwd_train(_p(x), _p(gamma), _p(beta), None, None, None,
                                       _p(running_mean), _p(running_var), _p(save_mean), _p(save_invstd), _p(partial), nsl,
                                       M, C, eps, momentum, 0, _dtype_code(x), _p(ws), ws.numel(), _stream()))
            defer.append(ws[:8 * C].view(torch.float32).clone())
            y = x.detach()                   # alias: the consumer reads the un-normalised tensor + the coefficients
        elif res_affine is not None:
            if residual is None:
                raise ValueError('batch_norm: res_affine without a residual operand')
            y = torch.empty_like(x, memory_format=CL)
            check(lib.cnk_bn_fwd_train_res_affine(_p(x), _p(gamma), _p(beta), _p(residual), _p(res_affine), _p(y), _p(mask),
                                                  _p(running_mean), _p(running_var), _p(save_mean), _p(save_invstd), _p(partial),
                                                  nsl, M, C, eps, momentum, int(relu), _dtype_code(x), _p(ws), ws.numel(),
                                                  _stream()))
        else:
            y = torch.empty_like(x, memory_format=CL)
            check(lib.cnk_bn_fwd_train(_p(x), _p(gamma), _p(beta), _p(residual), _p(y), _p(mask),
                                       _p(running_mean), _p(running_var), _p(save_mean), _p(save_invstd), _p(partial), nsl,
                                       M, C, eps, momentum, int(relu), _dtype_code(x), _p(ws), ws.numel(),
                                       _stream()))
        # the ReLU mask needs y only with a residual operand; otherwise the backward re-derives it
        ctx.save_for_backward(x, y if (relu and residual is not None and mask is None) else None,
                              mask, gamma, beta, save_mean, save_invstd)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.params = (gamma, beta)
        ctx.link = link if res_needs_grad else None
        ctx.rec = None
        ctx.fold = fold if (fold is not None and fold.ok and x_needs_grad and x.dtype == torch.bfloat16) else None
        if rec is not None and x_needs_grad and (mask is not None or residual is None):
            # what a consumer conv needs to run this layer's backward reduction in its dgrad epilogue
            rec.x, rec.mean, rec.invstd, rec.gamma, rec.beta = x, save_mean, save_invstd, gamma.detach(), beta.detach()
            rec.mask, rec.relu = mask, bool(relu)
            ctx.rec = rec
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, mask, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        # first pass already done by the consumer conv's dgrad epilogue?  Only if autograd hands back
        # exactly the tensor that conv wrote (no other gradient was accumulated on top of it)
        fused, ext_partial = None, None
        if ctx.rec is not None:
            fused, ctx.rec.fused = ctx.rec.fused, None
            ctx.rec.x = None
        if fused is not None and fused[0].data_ptr() == dy.data_ptr() and fused[0].shape == dy.shape \
                and dy.is_contiguous(memory_format=CL):
            ext_partial = fused[1]
        dy = _nhwc(dy)
        N, C, H, W = x.shape
        M = N * H * W
        dx = torch.empty_like(x, memory_format=CL)
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        FUSE_COUNTS['bn_bwd_fused' if ext_partial is not None else 'bn_bwd_unfused'] += 1
        if ext_partial is not None:
            return BatchNormActFn._backward_fused(ctx, lib, dy, x, gamma, beta, save_mean, save_invstd, dx,
                                                  want_res, ext_partial, M, C)
        # identity shortcut with a saved bit mask: the residual gradient dy*mask is NOT written here;
        # (dy, mask) is parked for the block's first conv, whose dgrad epilogue applies the mask
        defer_res = want_res and ctx.relu and mask is not None and ctx.link is not None \
            and ctx.link.arrived + 1 < ctx.link.count
        if defer_res:
            want_res = False
        if want_res and not ctx.relu:
            dres = dy                       # identity branch of a plain add: gradient passes through
            dres_out = None
        else:
            dres = torch.empty_like(x, memory_format=CL) if want_res else None
            dres_out = dres
        pg, pb = ctx.params
        gg, gb = _arena_grad(pg), _arena_grad(pb)
        direct = gg is not None and gb is not None
        if direct:
            dg_buf, db_buf, dgamma, dbeta = gg, gb, None, None
        else:
            dg_buf = dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
            db_buf = dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(bn_sizes(M, C, _dtype_code(x))[0], x.device)
        fold = ctx.fold
        if fold is not None and fold.ok and not ctx.relu and not ctx.has_res:
            # a BN without activation behind a foldable 1x1 conv (a projection shortcut): this call runs the reduction pass
            # only; the apply pass is folded into the conv's backward (FoldRecord), which receives dy itself
            coef = torch.empty(3 * C, dtype=torch.float32, device=x.device)
            check(lib.cnk_bn_bwd_coef(_p(dy), _p(x), None, 0, _p(gamma), _p(save_mean), _p(save_invstd), _p(dg_buf), _p(db_buf),
                                      int(direct), _p(coef), M, C, _dtype_code(x), _p(ws), ws.numel(), _stream()))
            fold.g, fold.y, fold.mean, fold.invstd, fold.coef = dy, x, save_mean, save_invstd, coef
            dx = dy
        else:
            check(lib.cnk_bn_bwd(_p(dy), _p(x), _p(y), _p(mask), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                 _p(dx), _p(dres_out), _p(dg_buf), _p(db_buf), int(direct), None, 0, M, C,
                                 int(ctx.relu), _dtype_code(x), _p(ws), ws.numel(), _stream()))
        if direct:
            _grad_written(pg)
            _grad_written(pb)
        if defer_res:
            ctx.link.contribute(lambda parked: (dy, mask))
            dres = None
        elif ctx.link is not None and dres is not None:
            # identity shortcut: park the residual gradient for the block's first conv, whose
            # dgrad epilogue adds it (falls back to a plain add if this backward is the last one)
            dres = ctx.link.contribute(lambda parked: dres if parked is None else dres + parked)
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None, None, None

    @staticmethod
    def _backward_fused(ctx, lib, dy, x, gamma, beta, save_mean, save_invstd, dx, want_res, ext_partial, M, C):
        """Second half of the backward when the consumer conv's dgrad epilogue has already masked dy
        and reduced it: dgamma/dbeta from the partials, then one apply pass.  The residual gradient
        is dy itself."""
        pg, pb = ctx.params
        gg, gb = _arena_grad(pg), _arena_grad(pb)
        direct = gg is not None and gb is not None
        if direct:
            dg_buf, db_buf, dgamma, dbeta = gg, gb, None, None
        else:
            dg_buf = dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
            db_buf = dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(bn_sizes(M, C, _dtype_code(x))[0], x.device)
        fold = ctx.fold
        if fold is not None and fold.ok:
            # the apply pass is folded into the backward of the 1x1 conv that produced x (FoldRecord): finalise only, and
            # hand dy itself on as that conv's incoming "gradient"
            coef = torch.empty(3 * C, dtype=torch.float32, device=x.device)
            check(lib.cnk_bn_bwd_coef(None, None, _p(ext_partial), ext_partial.shape[0], _p(gamma), _p(save_mean), _p(save_invstd),
                                      _p(dg_buf), _p(db_buf), int(direct), _p(coef), M, C, _dtype_code(x), _p(ws), ws.numel(),
                                      _stream()))
            fold.g, fold.y, fold.mean, fold.invstd, fold.coef = dy, x, save_mean, save_invstd, coef
            dx = dy
        else:
            check(lib.cnk_bn_bwd(_p(dy), _p(x), None, None, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                 _p(dx), None, _p(dg_buf), _p(db_buf), int(direct), _p(ext_partial),
                                 ext_partial.shape[0], M, C, int(ctx.relu), _dtype_code(x), _p(ws), ws.numel(),
                                 _stream()))
        if direct:
            _grad_written(pg)
            _grad_written(pb)
        dres = None
        if want_res:
            dres = dy
            if ctx.link is not None:
                dres = ctx.link.contribute(lambda parked: dy if parked is None else dy + parked)
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None, None, None


class _HalfCtx:
    """The autograd-context surface Conv2dFn / BatchNormActFn use (save_for_backward, saved_tensors, needs_input_grad, free
    attributes), for running their forward / backward bodies as the two halves of ONE node (ConvBnActFn)."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


class ConvBnActFn(torch.autograd.Function):
    """conv -> training-mode BatchNorm2d -> [ReLU] [+ residual] of one conv2d_block (basemodel.py:999-1015; resnet.py:191-194
    for the residual tail) as ONE autograd node.  The arithmetic is Conv2dFn's and BatchNormActFn's --
    their forward and backward bodies run here unchanged, on two half contexts, same kernels in the same order, so results are
    bit-identical to the two-node chain -- what goes away is host work: one Function.apply, one autograd node, one Module
    call chain less per block in each direction (ResNet-152's eager step issues 155 such blocks)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, running_mean, running_var, conv_args, bn_args):
        stride, padding, packed, link, bn_rec, groups, dilation, fold = conv_args
        eps, momentum, relu, res_link, rec, defer, res_affine = bn_args
        ctx.set_materialize_grads(False)
        c, b = _HalfCtx(), _HalfCtx()
        conv_needs_grad = x.requires_grad or weight.requires_grad
        y, partial = Conv2dFn.forward(c, x, weight, None, stride, padding, packed, True, link, bn_rec, groups, dilation, fold)
        out = BatchNormActFn.forward(b, y, gamma, beta, residual, running_mean, running_var, eps, momentum, relu, partial, res_link,
                                     rec, fold if (fold is not None and fold.ok) else None, defer, res_affine, conv_needs_grad)
        nc = len(c.saved_tensors)
        ctx.save_for_backward(*(tuple(c.saved_tensors) + tuple(b.saved_tensors)))
        c.saved_tensors = b.saved_tensors = None
        ctx.halves = (c, b, nc)
        if defer is not None:
            # the BN half returned an alias of the conv output (statistics only): hand on the conv output itself
            return y
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 9
        c, b, nc = ctx.halves
        st = ctx.saved_tensors
        c.saved_tensors, b.saved_tensors = st[:nc], st[nc:]
        need = ctx.needs_input_grad                      # x, weight, gamma, beta, residual, ...
        b.needs_input_grad = (need[0] or need[1], need[2], need[3], need[4]) + (False,) * 12
        gb = BatchNormActFn.backward(b, dout)
        c.needs_input_grad = (need[0], need[1], False) + (False,) * 10
        gc = Conv2dFn.backward(c, gb[0])
        c.saved_tensors = b.saved_tensors = None
        return gc[0], gc[1], gb[1], gb[2], gb[3], None, None, None, None


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps, residual=None, relu=False):
    """Eval-mode BN (valid_model, basemodel.py:311); inference only, no autograd graph."""
    _require_gpu(x)
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad):
        raise RuntimeError('convnets_amd: eval-mode batch norm is inference-only; wrap the call in '
                           'torch.no_grad() as the reference does (basemodel.py:304)')
    lib = _lib.load()
    x = _nhwc(x)
    if residual is not None:
        residual = _nhwc(residual)
    N, C, H, W = x.shape
    y = torch.empty_like(x, memory_format=CL)
    check(lib.cnk_bn_fwd_eval(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                              _p(residual), _p(y), N * H * W, C, eps, int(relu), _dtype_code(x),
                              _stream()))
    return y


POOL_KEEP = os.environ.get('CNK_POOL_KEEP', '1') != '0'      # (A/B switch for the kept arg-max inputs below)


class BatchNormActMaxPoolFn(torch.autograd.Function):
    """BatchNorm2d (training) -> [ReLU] -> MaxPool2d as ONE autograd node (the stems: resnet.py:88-89; VGG's stage
    ends).  Forward: the BN statistics, then a pooling pass that normalises the conv output on the fly
    (cnk_maxpool2d_fwd_bn) -- the full-size normalised tensor is neither written nor read.  Backward: the BN backward
    gathers the pooled gradient through the arg-max positions in both of its passes (cnk_bn_bwd_maxpool), ReLU mask
    re-derived from x; pools with more than 2 x 2 covering windows take max-pool backward + BN backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, partial, ksize, stride, padding):
        _require_gpu(x)
        lib = _lib.load()
        # (read before any layout copy made here: a copy made under no_grad drops the flag; the kept reduction also serves
        # dgamma / dbeta when x itself needs no gradient)
        needs_bwd = x.requires_grad or gamma.requires_grad or beta.requires_grad
        x = _nhwc(x)
        N, C, H, W = x.shape
        M = N * H * W
        save_mean = torch.empty(C, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(lib.cnk_bn_workspace_bytes(M, C), x.device)
        if partial is not None and partial.numel() == 0:
            partial = None
        nsl = 0 if partial is None else partial.shape[0]
        check(lib.cnk_bn_fwd_train(_p(x), _p(gamma), _p(beta), None, None, None, _p(running_mean), _p(running_var),
                                   _p(save_mean), _p(save_invstd), _p(partial), nsl, M, C, eps, momentum, int(relu),
                                   _dtype_code(x), _p(ws), ws.numel(), _stream()))
        Ho = (H + 2 * padding[0] - ksize[0]) // stride[0] + 1
        Wo = (W + 2 * padding[1] - ksize[1]) // stride[1] + 1
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=CL)
        idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
        scale, shift = ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(ws.data_ptr() + 4 * C)   # left there by the BN call
        # stride-2 pools whose backward gathers through <= 2 x 2 windows: the pooling pass also keeps the INPUT value at every
        # arg-max (a quarter of x), and the backward's reduction pass then reads that and the pooled gradient instead of
        # gathering over all of x (cnk_bn_bwd_maxpool_kept)
        keep = POOL_KEEP and needs_bwd and tuple(stride) == (2, 2) and ksize[0] <= 4 and ksize[1] <= 4
        xmax = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device) if keep else None
        if keep:
            check(lib.cnk_maxpool2d_fwd_bn_keep(_p(x), scale, shift, int(relu), _p(y), _p(idx), _p(xmax), N, H, W, C, ksize[0],
                                                ksize[1], stride[0], stride[1], padding[0], padding[1], _dtype_code(x), _stream()))
        else:
            check(lib.cnk_maxpool2d_fwd_bn(_p(x), scale, shift, int(relu), _p(y), _p(idx), N, H, W, C, ksize[0], ksize[1],
                                           stride[0], stride[1], padding[0], padding[1], _dtype_code(x), _stream()))
        ctx.save_for_backward(x, idx, gamma, beta, save_mean, save_invstd, xmax)
        ctx.relu, ctx.geom, ctx.params = bool(relu), (ksize, stride, padding), (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, idx, gamma, beta, save_mean, save_invstd, xmax = ctx.saved_tensors
        ksize, stride, padding = ctx.geom
        dy = _nhwc(dy)
        N, C, H, W = x.shape
        M = N * H * W
        dx = torch.empty_like(x, memory_format=CL)
        pg, pb = ctx.params
        gg, gb = _arena_grad(pg), _arena_grad(pb)
        direct = gg is not None and gb is not None
        if direct:
            dg_buf, db_buf, dgamma, dbeta = gg, gb, None, None
        else:
            dg_buf = dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
            db_buf = dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(lib.cnk_bn_workspace_bytes(M, C), x.device)
        if ksize[0] <= 2 * stride[0] and ksize[1] <= 2 * stride[1] and os.environ.get('CNK_BN_POOL_BWD', '1') != '0':
            # both BN passes gather the pooled gradient themselves: no full-size intermediate
            if xmax is not None:
                check(lib.cnk_bn_bwd_maxpool_kept(_p(dy), _p(idx), _p(x), _p(xmax), _p(gamma), _p(beta), _p(save_mean),
                                                  _p(save_invstd), _p(dx), _p(dg_buf), _p(db_buf), int(direct), N, H, W, C,
                                                  ksize[0], ksize[1], stride[0], stride[1], padding[0], padding[1],
                                                  int(ctx.relu), _dtype_code(x), _p(ws), ws.numel(), _stream()))
            else:
                check(lib.cnk_bn_bwd_maxpool(_p(dy), _p(idx), _p(x), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                             _p(dx), _p(dg_buf), _p(db_buf), int(direct), N, H, W, C, ksize[0], ksize[1],
                                             stride[0], stride[1], padding[0], padding[1], int(ctx.relu), _dtype_code(x),
                                             _p(ws), ws.numel(), _stream()))
        else:
            da = torch.empty_like(x, memory_format=CL)
            check(lib.cnk_maxpool2d_bwd(_p(dy), _p(idx), _p(da), N, H, W, C, ksize[0], ksize[1], stride[0], stride[1],
                                        padding[0], padding[1], _dtype_code(x), _stream()))
            check(lib.cnk_bn_bwd(_p(da), _p(x), None, None, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd), _p(dx),
                                 None, _p(dg_buf), _p(db_buf), int(direct), None, 0, M, C, int(ctx.relu), _dtype_code(x),
                                 _p(ws), ws.numel(), _stream()))
        if direct:
            _grad_written(pg)
            _grad_written(pb)
        return (dx, dgamma, dbeta) + (None,) * 9


class MaxPool2dFn(torch.autograd.Function):
    """nn.MaxPool2d (basemodel.py:1021)."""

    @staticmethod
    def forward(ctx, x, ksize, stride, padding):
        _require_gpu(x)
        lib = _lib.load()
        x_needs_grad = x.requires_grad
        x = _nhwc(x)
        N, C, H, W = x.shape
        Ho = (H + 2 * padding[0] - ksize[0]) // stride[0] + 1
        Wo = (W + 2 * padding[1] - ksize[1]) // stride[1] + 1
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=CL)
        idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device) if x_needs_grad else None
        check(lib.cnk_maxpool2d_fwd(_p(x), _p(y), _p(idx), N, H, W, C, ksize[0], ksize[1], stride[0],
                                    stride[1], padding[0], padding[1], _dtype_code(x), _stream()))
        ctx.save_for_backward(idx)
        ctx.geom = (ksize, stride, padding, tuple(x.shape), x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        ksize, stride, padding, shape, dt = ctx.geom
        dy = _nhwc(dy)
        N, C, H, W = shape
        dx = torch.empty(shape, dtype=dt, device=dy.device, memory_format=CL)
        check(lib.cnk_maxpool2d_bwd(_p(dy), _p(idx), _p(dx), N, H, W, C, ksize[0], ksize[1], stride[0],
                                    stride[1], padding[0], padding[1], _dtype_code(dy), _stream()))
        return dx, None, None, None


class ReLUFn(torch.autograd.Function):
    """nn.ReLU / F.relu as a stand-alone op (basemodel.py:1012; resnet.py:154,194)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        lib = _lib.load()
        x = x if (x.is_contiguous() or x.is_contiguous(memory_format=CL)) else x.contiguous()
        y = torch.empty_like(x)
        check(lib.cnk_relu_fwd(_p(x), _p(y), x.numel(), _dtype_code(x), _stream()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        if dy.stride() != y.stride():
            dy = dy.contiguous(memory_format=CL) if y.is_contiguous(memory_format=CL) and y.dim() == 4 \
                else dy.contiguous()
        dx = torch.empty_like(y)
        check(lib.cnk_relu_bwd(_p(dy), _p(y), _p(dx), y.numel(), _dtype_code(y), _stream()))
        return dx


class AddReLUFn(torch.autograd.Function):
    """`output += identity; output = F.relu(output)` (resnet.py:151-154,191-194) for blocks whose
    last conv has no BN to fuse into."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a)
        lib = _lib.load()
        a, b = _nhwc(a), _nhwc(b)
        y = torch.empty_like(a, memory_format=CL)
        check(lib.cnk_add_relu_fwd(_p(a), _p(b), _p(y), a.numel(), _dtype_code(a), _stream()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        dy = _nhwc(dy)
        dx = torch.empty_like(y, memory_format=CL)
        check(lib.cnk_relu_bwd(_p(dy), _p(y), _p(dx), y.numel(), _dtype_code(y), _stream()))
        return dx, dx


class GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) (basemodel.py:1039); returns fp32 [N,C,1,1] for the fp32 head."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        lib = _lib.load()
        x = _nhwc(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, 1, 1), dtype=torch.float32, device=x.device)
        vec = 8 if x.dtype == torch.bfloat16 else 4
        if H * W >= 64 and C % vec == 0:
            # large maps (the squeeze step of an SE unit): the chunked streaming reduction of csrc/sk.hip, one path
            nws = lib.cnk_sk_workspace_bytes(N, H * W, C, 1)
            ws = workspace(nws, x.device)
            check(lib.cnk_sk_pool_fwd(_ptr_table([x]), 1, _p(y), N, H * W, C, _dtype_code(x), _p(ws), nws, _stream()))
        else:
            check(lib.cnk_gap_fwd(_p(x), _p(y), N, H * W, C, _dtype_code(x), _stream()))
        ctx.shape, ctx.dt = x.shape, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, C, H, W = ctx.shape
        dy = dy.contiguous().float()
        dx = torch.empty(ctx.shape, dtype=ctx.dt, device=dy.device, memory_format=CL)
        check(lib.cnk_gap_bwd(_p(dy), _p(dx), N, H * W, C, CNK_BF16 if ctx.dt == torch.bfloat16 else CNK_F32,
                              _stream()))
        return dx


def nchw_to_nhwc(x, dtype):
    """Model-boundary hand-off: the reference feeds NCHW fp32 batches (basemodel.py:185-188)."""
    _require_gpu(x)
    lib = _lib.load()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    N, C, H, W = x.shape
    y = torch.empty((N, C, H, W), dtype=dtype, device=x.device, memory_format=CL)
    check(lib.cnk_nchw_to_nhwc(_p(x), _p(y), N, C, H, W, CNK_BF16 if dtype == torch.bfloat16 else CNK_F32,
                               _stream()))
    return y


# ---- classifier head ---------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """nn.Linear (resnet.py:106; vggnet.py:83-92) on the fp32 MFMA GEMM of csrc/head.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x)
        lib = _lib.load()
        if x.dim() != 2 or x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise ValueError('linear: expected fp32 x[B][K], got %s %s' % (tuple(x.shape), x.dtype))
        x = x.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        B, K = x.shape
        N = w.shape[0]
        if w.shape[1] != K:
            raise ValueError('linear: x has %d features, weight expects %d' % (K, w.shape[1]))
        y = torch.empty((B, N), dtype=torch.float32, device=x.device)
        ws = workspace(lib.cnk_linear_workspace_bytes(PASS_FWD, B, K, N), x.device)
        check(lib.cnk_linear_fwd(_p(x), _p(w), _p(bias), _p(y), B, K, N, _p(ws), ws.numel(), _stream()))
        ctx.save_for_backward(x, w)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        weight, bias = ctx.params
        dy = dy.contiguous()
        B, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, K), dtype=torch.float32, device=x.device)
            ws = workspace(lib.cnk_linear_workspace_bytes(PASS_DGRAD, B, K, N), x.device)
            check(lib.cnk_linear_dgrad(_p(dy), _p(w), _p(dx), B, K, N, _p(ws), ws.numel(), _stream()))
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            gw = _arena_grad(weight) if weight.is_contiguous() else None
            gb = _arena_grad(bias) if bias is not None else None
            direct = gw is not None and (bias is None or gb is not None)
            if direct:
                dw_buf, db_buf = gw, gb
            else:
                dw_buf = dw = torch.empty_like(w)
                db_buf = db = torch.empty(N, dtype=torch.float32, device=x.device) if bias is not None else None
            ws = workspace(lib.cnk_linear_workspace_bytes(PASS_WGRAD, B, K, N), x.device)
            check(lib.cnk_linear_wgrad(_p(dy), _p(x), _p(dw_buf), _p(db_buf), int(direct), B, K, N,
                                       _p(ws), ws.numel(), _stream()))
            if direct:
                _grad_written(weight)
                if bias is not None:
                    _grad_written(bias)
        return dx, dw, db


_dropout_calls = [0]


class CaptureScalars:
    """Step-varying scalars of a hipGraph-captured training step (convnets_amd/graph.py).  A captured launch
    replays with the arguments it was captured with, so what changes from step to step -- Adam's two bias
    corrections and the dropout call counter -- lives in a 16-byte device record that the host refreshes
    (pinned ring + async copy on the replay stream) before every replay: [bc1 f32][bc2_sqrt f32][dropout base u64]."""

    RING = 64

    def __init__(self, device):
        self.dev = torch.zeros(16, dtype=torch.uint8, device=device)
        self.ring = torch.zeros(self.RING, 16, dtype=torch.uint8).pin_memory()
        self.events = [None] * self.RING
        self.slot = 0
        self.dropout_base = 0            # _dropout_calls at the start of the captured step
        self.dropout_calls = 0           # dropout launches inside one step

    @property
    def bias_ptr(self):
        return ctypes.c_void_p(self.dev.data_ptr())

    @property
    def dropout_ptr(self):
        return ctypes.c_void_p(self.dev.data_ptr() + 8)

    def push(self, bc, dropout_base):
        """Queue the record of the next replay on the current stream."""
        import struct
        i = self.slot
        self.slot = (i + 1) % self.RING
        if self.events[i] is not None:
            self.events[i].synchronize()          # the copy that last read this pinned slot (RING replays ago)
        self.ring[i].copy_(torch.frombuffer(bytearray(struct.pack('<ffQ', bc[0], bc[1], dropout_base)), dtype=torch.uint8))
        self.dev.copy_(self.ring[i], non_blocking=True)
        ev = self.events[i] or torch.cuda.Event()
        ev.record()
        self.events[i] = ev


CAPTURE = [None]          # the CaptureScalars of the step being captured, else None



class DropoutFn(torch.autograd.Function):
    """nn.Dropout, training mode (resnet.py:105; feature maps: densenet.py, squeezenet.py), fp32 or bf16, any dense
    layout.  Keep pattern = hash(torch.initial_seed(), call counter, position in memory): reproducible under
    torch.manual_seed, not torch's Philox stream.  The mask is saved bit-packed."""

    @staticmethod
    def forward(ctx, x, p):
        _require_gpu(x)
        lib = _lib.load()
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('dropout: fp32 or bf16 tensors')
        x = x if (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=CL))) else x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty((x.numel() + 7) // 8, dtype=torch.uint8, device=x.device)
        _dropout_calls[0] += 1
        cap = CAPTURE[0]
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if cap is None:
            check(lib.cnk_dropout_fwd(_p(x), _p(y), _p(mask), x.numel(), float(p), seed, _dropout_calls[0], None,
                                      _dtype_code(x), _stream()))
        else:        # captured step: counter = (calls before this step, from device memory) + index inside the step
            cap.dropout_calls += 1
            check(lib.cnk_dropout_fwd(_p(x), _p(y), _p(mask), x.numel(), float(p), seed, cap.dropout_calls,
                                      cap.dropout_ptr, _dtype_code(x), _stream()))
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        ctx.fmt = CL if (x.dim() == 4 and not x.is_contiguous()) else torch.contiguous_format
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=ctx.fmt)
        dx = torch.empty_like(dy)
        check(lib.cnk_dropout_bwd(_p(dy), _p(mask), _p(dx), dy.numel(), ctx.p, _dtype_code(dy), _stream()))
        return dx, None


class SoftmaxCrossEntropySumFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(reduction='sum') (basemodel.py:46): loss and d(loss)/d(logits) in one pass.
    Also returns the per-sample argmax (non-differentiable) so score() needs no second pass."""

    @staticmethod
    def forward(ctx, logits, target):
        _require_gpu(logits)
        lib = _lib.load()
        if logits.dim() != 2 or logits.dtype != torch.float32:
            raise ValueError('cross entropy: expected fp32 logits[B][N], got %s %s' % (tuple(logits.shape), logits.dtype))
        if target.dtype != torch.int64 or target.shape != logits.shape[:1]:
            raise ValueError('cross entropy: expected int64 class targets of shape [B]')
        logits = logits.contiguous()
        target = target.contiguous()
        B, N = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        need = logits.requires_grad
        dlogits = torch.empty_like(logits) if need else None
        pred = torch.empty(B, dtype=torch.int32, device=logits.device)
        ws = workspace(B * 4, logits.device)
        check(lib.cnk_softmax_ce_sum(_p(logits), _p(target), _p(loss), _p(dlogits), _p(pred), B, N,
                                     _p(ws), ws.numel(), _stream()))
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)
        return loss, pred

    @staticmethod
    def backward(ctx, dloss, _dpred=None):
        (dlogits,) = ctx.saved_tensors
        if dloss is None or dlogits is None:
            return None, None
        return dlogits * dloss, None


# ---- inference: BatchNorm folded into the conv, activation in the conv epilogue -----
class FoldedConv:
    """Eval-mode conv2d_block (basemodel.py:999-1015 under model.eval()): the BN affine map of the
    running statistics is folded into the conv -- w' = w * gamma/sqrt(var+eps) per output channel,
    bias' = beta - mean*scale (+ conv bias*scale) -- and ReLU / the residual add run in the conv
    epilogue (cnk_conv2d_fwd_act).  The folded image is cached until the weights or the running
    statistics change (`stamp`)."""

    def __init__(self):
        self.stamp = None
        self.w_img = self.bias = None

    def run(self, x, conv_weight, conv_bias, bn, stride, padding, residual, relu, stamp, groups=1, dilation=1):
        _require_gpu(x)
        lib = _lib.load()
        x = _nhwc(x)
        if groups > 1 and groups == x.shape[1]:
            return None                                   # depthwise: own kernels, not folded
        desc = make_desc(x.shape, conv_weight.shape, stride, padding, _dtype_code(x), groups, dilation)
        if not lib.cnk_conv2d_fwd_act_supported(ctypes.byref(desc)):
            return None
        key = (stamp, desc.dtype, x.device)
        if self.stamp != key:
            with torch.no_grad():
                w = conv_weight.detach().float()
                if bn is not None:
                    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias.detach() - bn.running_mean * scale
                    if conv_bias is not None:
                        shift = shift + conv_bias.detach() * scale
                    w = w * scale.view(-1, 1, 1, 1)
                else:
                    shift = conv_bias.detach().float() if conv_bias is not None else None
                w = w.contiguous()
                self.w_img = torch.empty(lib.cnk_conv2d_packed_bytes(ctypes.byref(desc), PASS_FWD),
                                         dtype=torch.uint8, device=x.device)
                check(lib.cnk_conv2d_pack_weights(ctypes.byref(desc), _p(w), _p(self.w_img), None, _stream()))
                self.bias = shift.contiguous() if shift is not None else None
            self.stamp = key
        if residual is not None:
            residual = _nhwc(residual)
            if residual.dtype != x.dtype:
                raise ValueError('conv epilogue: residual dtype must match the activations')
        Ho, Wo = ctypes.c_int32(), ctypes.c_int32()
        lib.cnk_conv2d_out_hw(ctypes.byref(desc), ctypes.byref(Ho), ctypes.byref(Wo))
        y = torch.empty((desc.N, desc.Cout, Ho.value, Wo.value), dtype=x.dtype, device=x.device, memory_format=CL)
        if residual is not None and residual.shape != y.shape:
            raise ValueError('conv epilogue: residual shape %s != output shape %s' % (tuple(residual.shape), tuple(y.shape)))
        check(lib.cnk_conv2d_fwd_act(ctypes.byref(desc), _p(x), _p(self.w_img), _p(self.bias), _p(residual),
                                     int(relu), _p(y), _stream()))
        return y


# ---- depthwise convolution ---------------------------------------------------------
class DepthwiseConv2dFn(torch.autograd.Function):
    """nn.Conv2d(C, C, groups=C) -- conv2d_depthwise (basemodel.py:981-997) -- on csrc/dwconv.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        _require_gpu(x)
        lib = _lib.load()
        x = _nhwc(x)
        w = weight if weight.is_contiguous() else weight.contiguous()
        C = x.shape[1]
        if w.shape[0] != C or w.shape[1] != 1:
            raise ValueError('depthwise conv: weight %s does not match %d input channels' % (tuple(w.shape), C))
        desc = make_desc(x.shape, (C, C, w.shape[2], w.shape[3]), stride, padding, _dtype_code(x))
        if not lib.cnk_dwconv2d_supported(ctypes.byref(desc)):
            raise NotImplementedError('depthwise conv: C must be a multiple of %d and kh*kw <= 25 (got C=%d, %dx%d)'
                                      % (8 if x.dtype == torch.bfloat16 else 4, C, w.shape[2], w.shape[3]))
        Ho, Wo = ctypes.c_int32(), ctypes.c_int32()
        lib.cnk_conv2d_out_hw(ctypes.byref(desc), ctypes.byref(Ho), ctypes.byref(Wo))
        y = torch.empty((desc.N, C, Ho.value, Wo.value), dtype=x.dtype, device=x.device, memory_format=CL)
        check(lib.cnk_dwconv2d_fwd(ctypes.byref(desc), _p(x), _p(w), _p(bias), _p(y), _stream()))
        ctx.save_for_backward(x, w)
        ctx.desc, ctx.params = desc, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        weight, bias = ctx.params
        desc = ctx.desc
        dy = _nhwc(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=CL)
            check(lib.cnk_dwconv2d_dgrad(ctypes.byref(desc), _p(dy), _p(w), _p(dx), _stream()))
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            gw = _arena_grad(weight) if weight.is_contiguous() else None
            gb = _arena_grad(bias) if bias is not None else None
            direct = gw is not None and (bias is None or gb is not None)
            if direct:
                dw_buf, db_buf = gw, gb
            else:
                dw_buf = dw = torch.empty(w.shape, dtype=torch.float32, device=w.device)
                db_buf = db = torch.empty(w.shape[0], dtype=torch.float32, device=w.device) if bias is not None else None
            ws = workspace(lib.cnk_dwconv2d_workspace_bytes(ctypes.byref(desc)), x.device)
            check(lib.cnk_dwconv2d_wgrad(ctypes.byref(desc), _p(dy), _p(x), _p(dw_buf), _p(db_buf), int(direct),
                                         _p(ws), ws.numel(), _stream()))
            if direct:
                _grad_written(weight)
                if bias is not None:
                    _grad_written(bias)
        return dx, dw, db, None, None


class AvgPool2dFn(torch.autograd.Function):
    """nn.AvgPool2d (avgpool2d, basemodel.py:1026-1033): count_include_pad=True, floor mode."""

    @staticmethod
    def forward(ctx, x, ksize, stride, padding):
        _require_gpu(x)
        lib = _lib.load()
        x = _nhwc(x)
        N, C, H, W = x.shape
        Ho = (H + 2 * padding[0] - ksize[0]) // stride[0] + 1
        Wo = (W + 2 * padding[1] - ksize[1]) // stride[1] + 1
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=CL)
        check(lib.cnk_avgpool2d_fwd(_p(x), _p(y), N, H, W, C, ksize[0], ksize[1], stride[0], stride[1],
                                    padding[0], padding[1], _dtype_code(x), _stream()))
        ctx.geom = (ksize, stride, padding, tuple(x.shape), x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        ksize, stride, padding, shape, dt = ctx.geom
        dy = _nhwc(dy)
        N, C, H, W = shape
        dx = torch.empty(shape, dtype=dt, device=dy.device, memory_format=CL)
        check(lib.cnk_avgpool2d_bwd(_p(dy), _p(dx), N, H, W, C, ksize[0], ksize[1], stride[0], stride[1],
                                    padding[0], padding[1], _dtype_code(dy), _stream()))
        return dx, None, None, None


# ---- squeeze-and-excitation unit ----------------------------------------------------
class SigmoidFn(torch.autograd.Function):
    """nn.Sigmoid on the fp32 excitation descriptor (senet.py:236)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(_lib.load().cnk_sigmoid_fwd(_p(x), _p(y), x.numel(), _stream()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(_lib.load().cnk_sigmoid_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream()))
        return dx


class ChannelScaleFn(torch.autograd.Function):
    """x * gate.expand_as(x) (senet.py:252): NHWC activations scaled per (sample, channel) by the fp32 gate [N, C].
    link: the SKLink shared with the SKPoolFn that squeezed the same x (layers.SEUnit): this backward then produces
    only dgate and leaves (dy, gate) for the pool backward, which writes dx = dy * gate + dpooled / HW in one pass
    instead of two gradient tensors plus autograd's add."""

    @staticmethod
    def forward(ctx, x, gate, link=None):
        _require_gpu(x)
        lib = _lib.load()
        x = _nhwc(x)
        gate = gate.contiguous()
        N, C, H, W = x.shape
        if gate.shape != (N, C) or gate.dtype != torch.float32:
            raise ValueError('channel scale: gate must be fp32 [N, C]')
        y = torch.empty_like(x, memory_format=CL)
        check(lib.cnk_channel_scale_fwd(_p(x), _p(gate), _p(y), N, H * W, C, _dtype_code(x), _stream()))
        ctx.save_for_backward(x, gate)
        ctx.link = link
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gate = ctx.saved_tensors
        dy = _nhwc(dy)
        N, C, H, W = x.shape
        link = ctx.link
        linked = link is not None and link.pool_pending and ctx.needs_input_grad[0]
        dx = None if linked else torch.empty_like(x, memory_format=CL)
        dgate = torch.empty_like(gate)
        nws = lib.cnk_channel_scale_workspace_bytes(N, H * W, C, _dtype_code(x))
        ws = workspace(nws, x.device) if nws else None
        check(lib.cnk_channel_scale_bwd(_p(dy), _p(x), _p(gate), _p(dx), _p(dgate), N, H * W, C, _dtype_code(x),
                                        _p(ws), nws, _stream()))
        if linked:
            link.stash = (dy, gate)                    # SKPoolFn.backward writes the gradient of x
        return dx, dgate, None


# ---- selective-kernel unit (sknet.py:172-240) ---------------------------------------------------------
def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class SKLink:
    """Joins the two backward passes of one SKConv.  Each feature map U_i is read twice in forward (by the fuse
    sum and by the select sum).  The select backward runs first (its output is the block's activation); it hands
    (dout, attn) to the pool backward, which runs last (its gradient comes back through descriptor + attentions)
    and writes dU_i = dout * a_i + dpooled / HW in ONE pass over the maps instead of two passes plus an add."""

    def __init__(self):
        self.pool_pending = False
        self.stash = None


class SKPoolFn(torch.autograd.Function):
    """torch.sum(feature_maps, dim=1) + AdaptiveAvgPool2d(1) (sknet.py:227-228) -> fp32 [N, C, 1, 1]."""

    @staticmethod
    def forward(ctx, link, *maps):
        _require_gpu(maps[0])
        lib = _lib.load()
        maps = [_nhwc(u) for u in maps]
        N, C, H, W = maps[0].shape
        if any(u.shape != maps[0].shape or u.dtype != maps[0].dtype for u in maps):
            raise ValueError('sk pool: the paths must produce one shape and dtype')
        pooled = torch.empty((N, C, 1, 1), dtype=torch.float32, device=maps[0].device)
        nws = lib.cnk_sk_workspace_bytes(N, H * W, C, 1)
        ws = workspace(nws, pooled.device)
        check(lib.cnk_sk_pool_fwd(_ptr_table(maps), len(maps), _p(pooled), N, H * W, C, _dtype_code(maps[0]), _p(ws), nws,
                                  _stream()))
        ctx.link, ctx.shape, ctx.dt, ctx.P = link, maps[0].shape, maps[0].dtype, len(maps)
        if link is not None:
            link.pool_pending = any(ctx.needs_input_grad[1:])
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        lib = _lib.load()
        N, C, H, W = ctx.shape
        dpooled = dpooled.contiguous().float()
        link = ctx.link
        dout, attn = (None, None)
        if link is not None:
            link.pool_pending = False
            if link.stash is not None:
                dout, attn = link.stash
                link.stash = None
        du = [torch.empty(ctx.shape, dtype=ctx.dt, device=dpooled.device, memory_format=CL) for _ in range(ctx.P)]
        check(lib.cnk_sk_maps_bwd(_p(dout), _p(attn), _p(dpooled), _ptr_table(du), ctx.P, N, H * W, C,
                                  CNK_BF16 if ctx.dt == torch.bfloat16 else CNK_F32, _stream()))
        return (None,) + tuple(du)


class SKSelectFn(torch.autograd.Function):
    """softmax over the paths of the attention logits, then the attention-weighted sum of the feature maps
    (sknet.py:231-237).  Arguments: link, P, then the P maps, then the P fp32 logits [N, C, 1, 1]."""

    @staticmethod
    def forward(ctx, link, P, *args):
        lib = _lib.load()
        maps = [_nhwc(u) for u in args[:P]]
        logits = [l.contiguous() for l in args[P:]]
        _require_gpu(maps[0])
        N, C, H, W = maps[0].shape
        if len(logits) != P or any(l.dtype != torch.float32 or l.numel() != N * C for l in logits):
            raise ValueError('sk select: one fp32 [N, C, 1, 1] logit tensor per path')
        attn = torch.empty((P, N, C), dtype=torch.float32, device=maps[0].device)
        out = torch.empty_like(maps[0], memory_format=CL)
        check(lib.cnk_sk_select_fwd(_ptr_table(maps), _ptr_table(logits), P, _p(attn), _p(out), N, H * W, C,
                                    _dtype_code(out), _stream()))
        ctx.save_for_backward(attn, *maps)
        ctx.link, ctx.P, ctx.lshape = link, P, logits[0].shape
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        attn, maps = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        P, link = ctx.P, ctx.link
        dout = _nhwc(dout)
        N, C, H, W = maps[0].shape
        dt = _dtype_code(maps[0])
        dlogits = [torch.empty(ctx.lshape, dtype=torch.float32, device=dout.device) for _ in range(P)]
        nws = lib.cnk_sk_workspace_bytes(N, H * W, C, P)
        ws = workspace(nws, dout.device)
        check(lib.cnk_sk_select_bwd_logits(_ptr_table(maps), _p(attn), _p(dout), _ptr_table(dlogits), P, N, H * W, C, dt,
                                           _p(ws), nws, _stream()))
        if link is not None and link.pool_pending and any(ctx.needs_input_grad[2:2 + P]):
            link.stash = (dout, attn)                  # SKPoolFn.backward writes the map gradients
            return (None, None) + (None,) * P + tuple(dlogits)
        du = [torch.empty_like(maps[0], memory_format=CL) for _ in range(P)]
        check(lib.cnk_sk_maps_bwd(_p(dout), _p(attn), None, _ptr_table(du), P, N, H * W, C, dt, _stream()))
        return (None, None) + tuple(du) + tuple(dlogits)


def affine_fixed_coeffs(angle, scale, shear_x, width, height, translate=(0.0, 0.0), shear_y=0.0):
    """The six 16.16 fixed-point coefficients PIL's nearest-neighbour Image.transform(AFFINE) uses for the inverse
    matrix torchvision's RandomAffine builds from one draw (angles in degrees) -- cnk_affine_fixed_coeffs."""
    out = (ctypes.c_int32 * 6)()
    check(_lib.load().cnk_affine_fixed_coeffs(float(angle), float(translate[0]), float(translate[1]), float(scale),
                                              float(shear_x), float(shear_y), int(width), int(height), out))
    return list(out)


def input_pipeline(images_u8, mean, std, dtype, crop_size=None, padding=0, crop_yx=None, flip=None, affine=None):
    """The reference's train/valid transforms (mngrdata.py:139-165) on the device in one pass: RandomCrop(crop_size,
    padding) + RandomHorizontalFlip + RandomAffine + ToTensor + Normalize(mean, std), uint8 [N, H, W, C] -> NHWC
    `dtype` tensor [N, C, Ho, Wo] (channels_last) that the model forward takes as is.  crop_yx (int32 [N, 2]), flip
    (uint8/bool [N]) and affine (float [N, 3]: angle in degrees, scale, x-shear in degrees -- what
    RandomAffine(degrees, shear, scale).get_params draws) are the caller's random draws; None = centre crop / no
    flip / no affine.  No autograd (data)."""
    _require_gpu(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
        raise ValueError('input_pipeline: expected uint8 images [N, H, W, C]')
    x = images_u8.contiguous()
    N, H, W, C = x.shape
    Ho, Wo = (H, W) if crop_size is None else ((crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size))
    dev = x.device
    mean_t = torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous()
    std_t = torch.as_tensor(std, dtype=torch.float32, device=dev).contiguous()
    if crop_yx is not None:
        crop_yx = crop_yx.to(device=dev, dtype=torch.int32).contiguous()
    if flip is not None:
        flip = flip.to(device=dev, dtype=torch.uint8).contiguous()
    fix = None
    if affine is not None:
        draws = torch.as_tensor(affine, dtype=torch.float64).reshape(-1, 3).tolist()
        if len(draws) != N:
            raise ValueError('input_pipeline: affine needs one (angle, scale, shear_x) row per image')
        fix = torch.tensor([affine_fixed_coeffs(a, sc, sh, Wo, Ho) for a, sc, sh in draws], dtype=torch.int32).to(dev)
    y = torch.empty((N, C, Ho, Wo), dtype=dtype, device=dev, memory_format=CL)
    check(_lib.load().cnk_input_pipeline_affine(_p(x), _p(y), _p(crop_yx), _p(flip), _p(fix), _p(mean_t), _p(std_t), N, H,
                                                W, C, Ho, Wo, int(padding), _dtype_code(y), _stream()))
    return y


# ---- general grouped conv through the dense kernels; channel shuffle (ShuffleNet) ------
class GroupedDenseWeightFn(torch.autograd.Function):
    """[Cout, Cin/g, kh, kw] -> dense block-diagonal [Cout, Cin, kh, kw]; backward extracts the diagonal blocks."""

    @staticmethod
    def forward(ctx, weight, groups):
        w = weight.contiguous()
        Cout, gw, kh, kw = w.shape
        wd = torch.empty((Cout, gw * groups, kh, kw), dtype=torch.float32, device=w.device)
        check(_lib.load().cnk_grouped_expand(_p(w), _p(wd), Cout, gw * groups, groups, kh * kw, _stream()))
        ctx.geom = (Cout, gw * groups, groups, kh, kw)
        return wd

    @staticmethod
    def backward(ctx, dwd):
        Cout, Cin, groups, kh, kw = ctx.geom
        dwd = dwd.contiguous()
        dw = torch.empty((Cout, Cin // groups, kh, kw), dtype=torch.float32, device=dwd.device)
        check(_lib.load().cnk_grouped_extract(_p(dwd), _p(dw), Cout, Cin, groups, kh * kw, _stream()))
        return dw, None


class ChannelShuffleFn(torch.autograd.Function):
    """view(N, g, C/g, H, W).transpose(1, 2).contiguous().view(N, C, H, W) (shufflenet_v1.py:173-186) on NHWC."""

    @staticmethod
    def forward(ctx, x, groups):
        _require_gpu(x)
        x = _nhwc(x)
        N, C, H, W = x.shape
        y = torch.empty_like(x, memory_format=CL)
        check(_lib.load().cnk_channel_shuffle(_p(x), _p(y), N * H * W, C, groups, _dtype_code(x), _stream()))
        ctx.groups, ctx.C = groups, C
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _nhwc(dy)
        N, C, H, W = dy.shape
        dx = torch.empty_like(dy, memory_format=CL)
        check(_lib.load().cnk_channel_shuffle(_p(dy), _p(dx), N * H * W, C, C // ctx.groups, _dtype_code(dy), _stream()))
        return dx, None
