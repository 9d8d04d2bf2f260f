"""CPU suite: the C-ABI library loads and exports every symbol include/cnk.h declares (no compute
calls without a GPU), host-side planning entry points are consistent, and the Python mirror of the
reference's factory behaves like the reference's (names, kwargs, shape tracker, state_dict keys,
error behaviour).  The product path must fail LOUDLY without a GPU -- never fall back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = np.load(os.path.join(ROOT, 'tests', 'golden', 'models.npz'))


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'cnk.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cnk_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from convnets_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), 'libcnk.so does not export %s' % s
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes table and header disagree'
    assert lib.cnk_version() >= 100


def test_host_side_planning_queries_run_without_gpu():
    from convnets_amd import _lib
    lib = _lib.load()
    D = _lib.ConvDesc
    # (N,H,W,Cin,Cout,kh,kw,sh,sw,ph,pw,dtype)
    d = D(256, 56, 56, 64, 64, 3, 3, 1, 1, 1, 1, _lib.CNK_BF16)
    Ho, Wo = ctypes.c_int32(), ctypes.c_int32()
    assert lib.cnk_conv2d_out_hw(ctypes.byref(d), ctypes.byref(Ho), ctypes.byref(Wo)) == 0
    assert (Ho.value, Wo.value) == (56, 56)
    n = 64 * 64 * 9
    assert lib.cnk_conv2d_packed_bytes(ctypes.byref(d), _lib.PASS_FWD) == n * 2      # bf16 image
    assert lib.cnk_conv2d_packed_bytes(ctypes.byref(d), _lib.PASS_DGRAD) == n * 2
    ws = lib.cnk_conv2d_workspace_bytes(ctypes.byref(d), _lib.PASS_WGRAD)
    assert ws >= n * 4 and ws % 4 == 0                                              # >= one fp32 slab
    stem = D(256, 224, 224, 3, 64, 7, 7, 2, 2, 3, 3, _lib.CNK_BF16)
    assert lib.cnk_conv2d_out_hw(ctypes.byref(stem), ctypes.byref(Ho), ctypes.byref(Wo)) == 0
    assert (Ho.value, Wo.value) == (112, 112)
    assert lib.cnk_conv2d_packed_bytes(ctypes.byref(stem), _lib.PASS_FWD) == 64 * 192 * 2   # 7 runs of 24 -> K' 192
    assert lib.cnk_bn_workspace_bytes(256 * 56 * 56, 64) > 0
    assert lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(d)) == 256 * 56 * 56 // 64    # one per 64-row wave tile (ws3x3)
    assert lib.cnk_conv2d_fwd_bn_partial_slices(ctypes.byref(stem)) == 256 * 112 // 56    # row-patch stem: one per workgroup (56 output rows: one resident round of 512)
    # argument errors come back as codes + text, never exit/abort
    bad = D(1, 4, 4, 8, 8, 9, 9, 1, 1, 0, 0, _lib.CNK_F32)
    rc = lib.cnk_conv2d_fwd(ctypes.byref(bad), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
                            ctypes.c_void_p(16), None, None, 0, None)
    assert rc != 0 and b'kernel larger' in lib.cnk_last_error()
    with pytest.raises(_lib.CnkError):
        _lib.check(rc)
    # round 4: which convs can take the folded BN backward, the image size, the smallest weight-gradient workspace
    c3 = D(256, 56, 56, 64, 256, 1, 1, 1, 1, 0, 0, _lib.CNK_BF16)                  # a bottleneck's last conv
    assert lib.cnk_conv2d_bnfold_supported(ctypes.byref(c3)) == 1
    assert lib.cnk_conv2d_bnfold_image_bytes(ctypes.byref(c3)) == 64 * (256 + 64) * 2
    assert lib.cnk_conv2d_bnfold_supported(ctypes.byref(D(256, 56, 56, 256, 512, 1, 1, 2, 2, 0, 0, _lib.CNK_BF16))) == 0    # stride 1 only (round 5)
    assert lib.cnk_conv2d_bnfold_supported(ctypes.byref(D(256, 56, 56, 64, 256, 1, 1, 1, 1, 0, 0, _lib.CNK_F32))) == 0      # bf16 only
    assert lib.cnk_conv2d_bnfold_supported(ctypes.byref(d)) == 0                                                        # 3x3
    assert lib.cnk_conv2d_bnfold_supported(ctypes.byref(D(8, 14, 14, 48, 96, 1, 1, 1, 1, 0, 0, _lib.CNK_BF16))) == 0        # off the 64 grid
    rc = lib.cnk_conv2d_bnfold_prepare(ctypes.byref(d), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                                       ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None)
    assert rc != 0 and b'cnk_conv2d_bnfold_supported' in lib.cnk_last_error()
    slab = 256 * 64 * 4
    assert lib.cnk_conv2d_wgrad_min_workspace_bytes(ctypes.byref(c3), 0) == slab
    assert lib.cnk_conv2d_wgrad_min_workspace_bytes(ctypes.byref(c3), 1) > slab                                            # + the bias sum's scratch
    assert lib.cnk_conv2d_wgrad_min_workspace_bytes(ctypes.byref(c3), 0) <= lib.cnk_conv2d_workspace_bytes(ctypes.byref(c3), _lib.PASS_WGRAD)


def test_ops_refuse_cpu_tensors_no_fallback():
    from convnets_amd import ops
    import convnets_amd as ca
    x = torch.randn(2, 64, 8, 8)
    for fn in (lambda: ops.ReLUFn.apply(x), lambda: ops.GlobalAvgPoolFn.apply(x),
               lambda: ops.MaxPool2dFn.apply(x, (2, 2), (2, 2), (0, 0)),
               lambda: ops.nchw_to_nhwc(x, torch.bfloat16),
               lambda: ops.Conv2dFn.apply(x, torch.randn(64, 64, 3, 3), None, (1, 1), (1, 1), ops.PackedWeights())):
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            fn()
    m = ca.ConvNet(ca.Settings(kind=0, input_size=(1, 28, 28), num_classes=10, batch_norm=False, device='cpu'))
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 1, 28, 28))
    with pytest.raises(RuntimeError, match='HIP path only'):
        m.init_optimizer()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'convnets_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            assert 'oracle' not in re.sub(r'""".*?"""', '', open(os.path.join(pkg, f)).read(), flags=re.S), f


def test_factory_returns_stock_subclasses_and_tracks_shapes():
    import convnets_amd as ca
    s = ca.Settings(kind=0, input_size=(3, 224, 224), num_classes=10, device='cpu')
    net = ca.MultiClassBaseModel(s)
    blk = net.conv2d_block(num_filters=64, kernel_size=7, stride=2, padding=3)
    assert isinstance(blk, nn.Sequential) and [type(m).__name__ for m in blk] == ['Conv2d', 'BatchNorm2d', 'ReLU']
    assert isinstance(blk[0], nn.Conv2d) and isinstance(blk[1], nn.BatchNorm2d) and isinstance(blk[2], nn.ReLU)
    assert blk[0].bias is None                                   # bias iff not batch_norm (basemodel.py:973)
    assert (net.in_channels, net.height, net.width) == (64, 112, 112)
    mp = net.maxpool2d(kernel_size=3, stride=2, padding=1)
    assert isinstance(mp, nn.MaxPool2d) and (net.height, net.width) == (56, 56)
    ds = net.conv2d_block(64, 256, set_output=False, activation=False, kernel_size=1, stride=2)
    assert len(ds) == 2 and (net.in_channels, net.height) == (64, 56)   # set_output=False leaves the tracker alone
    gap = net.adapt_avgpool2d(output_size=1)
    assert isinstance(gap, nn.AdaptiveAvgPool2d) and net.num_flat_features() == 64
    nobn = ca.MultiClassBaseModel(ca.Settings(kind=0, input_size=(1, 28, 28), num_classes=10, batch_norm=False, device='cpu'))
    c = nobn.conv2d(num_filters=32, kernel_size=3)
    assert c.bias is not None and (nobn.height, nobn.width) == (26, 26)
    dwc = net.conv2d_depthwise(kernel_size=3, padding=1, set_output=False)       # basemodel.py:981-997
    assert isinstance(dwc, nn.Conv2d) and dwc.groups == 64 and tuple(dwc.weight.shape) == (64, 1, 3, 3)
    dm = net.conv2d_depthwise(num_filters=128, kernel_size=3, set_output=False)   # channel multiplier 2: dense grouped path
    assert dm.groups == 64 and dm._dense_grouped and not dm._depthwise
    gc = net.conv2d(num_filters=64, kernel_size=3, groups=32, set_output=False)    # ResNeXt cardinality (resnext.py:82-96)
    assert gc.groups == 32 and tuple(gc.weight.shape) == (64, 2, 3, 3)
    gd = net.conv2d(in_channels=96, num_filters=48, kernel_size=1, groups=3, set_output=False)   # ShuffleNet-style: dense path
    assert gd._dense_grouped and tuple(gd.weight.shape) == (48, 32, 1, 1)
    before = (net.height, net.width)
    dc = net.conv2d(in_channels=64, num_filters=64, kernel_size=3, dilation=2, padding=2, set_output=False)
    assert dc._dil == 2 and (net.height, net.width) == before                    # sknet.py:190-193 dilated paths
    with pytest.raises(NotImplementedError):                                     # dilation only on the implicit-GEMM shapes
        net.conv2d(in_channels=3, num_filters=64, kernel_size=3, dilation=2)
    with pytest.raises(NotImplementedError):
        net.conv2d(in_channels=64, num_filters=64, kernel_size=3, dilation=(1, 2))
    with pytest.raises(TypeError):
        ca.Settings(kind=0, input_size=(1, 2, 2), num_classes=2, no_such_field=1)


@pytest.mark.parametrize('name,family,kind', [('resnet50', 'resnet', 50), ('resnet152', 'resnet', 152), ('vgg16', 'vgg', 16)])
def test_state_dict_interchange_with_reference(name, family, kind):
    """Same keys and shapes as the reference's classes (golden facts), so checkpoints interchange."""
    import convnets_amd as ca
    s = ca.Settings(kind=kind, input_size=(3, 224, 224), num_classes=1000, init_params=False, device='cpu')
    m = ca.MODELS[family](s)
    sd = m.state_dict()
    assert list(sd.keys()) == list(MODELS[f'{name}/sd_keys'])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(MODELS[f'{name}/sd_shapes'])
    assert sum(p.numel() for p in m.parameters()) == int(MODELS[f'{name}/n_params'])


def test_init_params_reproduces_reference_weights_and_deepcopy():
    import copy
    import convnets_amd as ca
    torch.manual_seed(21)
    m = ca.ResNet(ca.Settings(kind=26, input_size=(3, 32, 32), num_classes=10, device='cpu'))
    norms = np.array([float(p.detach().double().norm()) for p in m.parameters()])
    np.testing.assert_allclose(norms, MODELS['resnet26/init_norms'], rtol=1e-6)
    m2 = copy.deepcopy(m)                                         # basemodel.py:413 deep-copies state
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert m2.features[0][0]._packed is not m.features[0][0]._packed
    s = ca.Settings(kind=26, input_size=(3, 32, 32), num_classes=10, device='cpu')
    assert (s.batch_size, s.learning_rate, s.weight_decay, s.dropout_rate, s.seed) == (256, 0.01, 1e-4, 0.5, 21)
    assert s.compute_dtype == torch.bfloat16 and ca.Settings(kind=0, input_size=(1, 1, 1), num_classes=2, mixed_precision=False, device='cpu').compute_dtype == torch.float32


def test_affine_fixed_coeffs_host_helper_matches_oracle():
    """cnk_affine_fixed_coeffs is a pure host function (no GPU): torchvision's inverse affine matrix + Pillow's 16.16
    fixed-point conversion, identical integers to the restatement that tests/golden/affine.npz pins."""
    from convnets_amd import ops
    from oracle import transforms_ref as tf
    rng = np.random.RandomState(0)
    for _ in range(300):
        W, H = int(rng.randint(1, 300)), int(rng.randint(1, 300))
        a, sc, sh = rng.uniform(-180, 180), rng.uniform(0.2, 3), rng.uniform(-60, 60)
        m = tf.tv_inverse_affine_matrix([W * 0.5, H * 0.5], a, [0, 0], sc, [sh, 0.0])
        assert ops.affine_fixed_coeffs(a, sc, sh, W, H) == tf.pil_fixed_coeffs_abi_order(m)


def test_integration_snippet_matches_header():
    """INTEGRATION.md section C is the template a maintainer would copy: its ConvDesc must have the size of
    struct cnk_conv_desc in include/cnk.h, its cnk_conv2d_fwd argtypes the header's parameter count, and the example
    call must pass that many arguments (a short struct or a shifted stream argument would corrupt a launch silently)."""
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    block = md[md.index("```python\nimport ctypes\nlib = ctypes.CDLL('convnets_amd/libcnk.so')"):]
    block = block[len('```python\n'):block.index('\n```')]
    hdr = open(os.path.join(root, 'include', 'cnk.h')).read()
    struct = hdr[hdr.index('typedef struct cnk_conv_desc'):hdr.index('} cnk_conv_desc;')]
    n_fields = sum(len(re.sub(r'/\*.*?\*/', '', line).split(';')[0].split(',')) for line in struct.splitlines() if 'int32_t' in line)
    # the snippet's class definition and argtypes, executed without loading any library
    ns = {'ctypes': ctypes}
    cls_src = block[block.index('class ConvDesc'):block.index('# int cnk_conv2d_fwd')]
    exec(cls_src, ns)
    from convnets_amd import _lib
    assert ctypes.sizeof(ns['ConvDesc']) == 4 * n_fields == ctypes.sizeof(_lib.ConvDesc)
    assert [f[0] for f in ns['ConvDesc']._fields_] == [f[0] for f in _lib.ConvDesc._fields_]
    proto = re.search(r'int cnk_conv2d_fwd\((.*?)\);', hdr, re.S).group(1)
    n_params = len(proto.split(','))
    argtypes = eval(re.search(r'lib\.cnk_conv2d_fwd\.argtypes = (.*)', block).group(1), {'ctypes': ctypes, 'ConvDesc': ns['ConvDesc']})
    assert len(argtypes) == n_params == len(_lib.SIGNATURES['cnk_conv2d_fwd'][1])
    call = re.search(r'rc = lib\.cnk_conv2d_fwd\((.*)\)\s*#', block).group(1)
    assert len(call.split(',')) == n_params
    ctor = re.search(r'd = ConvDesc\((.*?)\)', block).group(1)
    assert len(ctor.split(',')) == n_fields


def test_committed_bench_line_keeps_the_contract():
    """The newest committed bench line (profiles/r05*_bench.json, written by `python bench.py` on an MI355X) carries every field
    of the bench contract with the agreed types: one JSON object, metric / config of BASELINE.json's headline, a numeric
    roofline.traffic, the cpu_baseline object (round 5: timed at the metric's batch), the whole-step numbers."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_bench.json')))
    assert files
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k, t in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                 ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str),
                 ('config', dict), ('roofline', dict), ('cpu_baseline', dict)):
        assert isinstance(d[k], t), (k, d.get(k))
    assert 'vs_baseline' in d and d['vs_baseline'] is None           # BASELINE.md holds no published number for this metric
    assert d['scaling'] == 'weak' and d['higher_is_better'] is True and d['dtype'] == 'bf16' and d['n_gpus'] == 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert 'images/sec' in d['metric'] and 'ResNet-50' in d['metric'], (d['metric'], base.get('metric'))
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is None or isinstance(r['traffic'], (int, float))
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str)
    assert abs(d['value'] - 256 * 1000.0 / d['ms_per_step']) / d['value'] < 1e-2     # value and ms_per_step describe the same run
    if os.path.basename(files[-1]) >= 'r05':
        assert c['batch'] in (256, 64) and (c['batch'] == 256 or c['batch_fallback'])
        sl = d['step_level']
        assert 0 < sl['step_tflops_frac'] < 1 and 0 < sl['step_hbm_frac'] < 1
        assert abs(sl['step_tflops'] - sl['step_algorithmic_tflop'] / d['ms_per_step'] * 1e3) / sl['step_tflops'] < 1e-2


def test_graphed_batch_dispatch_under_data_parallelism():
    """basemodel._graphed_batch (BASELINE config 5; the reference has one device, /root/reference/src/mngrdevice.py:18): under
    data parallelism the graph is built ONCE, collectively, at the first training batch; batches of that shape replay it,
    any other shape runs the eager step (same collectives as a replay) -- no rank can enter a second capture alone; without
    Settings.hipgraph_dp a multi-rank model trains eagerly."""
    import convnets_amd as ca
    from convnets_amd import graph
    from convnets_amd.basemodel import MultiClassBaseModel

    class Red:
        def __init__(self, world, force=False):
            self.world, self.force, self.group = world, force, 'grp'

    class FakeStep:
        def __init__(self):
            self.replays = 0

        def replay(self, batch):
            self.replays += 1
            return 'replay', batch[0].shape[0]

    class M(MultiClassBaseModel):
        def train_batch(self, batch):
            return 'eager', batch[0].shape[0]

    built = []
    orig = graph.GraphedTrainStep.build_collective

    def fake_build(model, batch, group=None, **kw):
        built.append((tuple(batch[0].shape), group))
        return FakeStep()
    graph.GraphedTrainStep.build_collective = staticmethod(fake_build)
    try:
        b16 = (torch.zeros(16, 3, 8, 8), torch.zeros(16, dtype=torch.int64))
        b8 = (torch.zeros(8, 3, 8, 8), torch.zeros(8, dtype=torch.int64))
        s = ca.Settings(kind=0, input_size=(3, 8, 8), num_classes=10, device='cpu', hipgraph=True, hipgraph_dp=True)
        m = M(s)
        m.grad_reducer = Red(world=2)
        assert m._graphed_batch(b16) == ('replay', 16) and built == [((16, 3, 8, 8), 'grp')]
        assert m._graphed_batch(b8) == ('eager', 8)                 # another shape: eager, no second construction
        assert m._graphed_batch(b16) == ('replay', 16) and len(built) == 1
        # a refused construction (None on every rank): eager from then on, never retried
        graph.GraphedTrainStep.build_collective = staticmethod(lambda *a, **k: built.append('refused') or None)
        m2 = M(s)
        m2.grad_reducer = Red(world=2)
        assert m2._graphed_batch(b16) == ('eager', 16) and m2._graphed_batch(b16) == ('eager', 16)
        assert built.count('refused') == 1
        # the flag off: multi-rank models train eagerly and never construct
        s_off = ca.Settings(kind=0, input_size=(3, 8, 8), num_classes=10, device='cpu', hipgraph=True, hipgraph_dp=False)
        m3 = M(s_off)
        m3.grad_reducer = Red(world=2)
        n = len(built)
        assert m3._graphed_batch(b16) == ('eager', 16) and len(built) == n
        # one rank with the collective forced (the single-GPU rehearsal of the path) takes the collective route, group None
        graph.GraphedTrainStep.build_collective = staticmethod(fake_build)
        m4 = M(s_off)
        m4.grad_reducer = Red(world=1, force=True)
        assert m4._graphed_batch(b16) == ('replay', 16) and built[-1] == ((16, 3, 8, 8), None)
    finally:
        graph.GraphedTrainStep.build_collective = orig
    assert ca.Settings(kind=0, input_size=(3, 8, 8), num_classes=10, device='cpu').hipgraph_dp in (False, True)
