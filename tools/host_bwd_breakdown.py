"""Where the host time of the eager BACKWARD goes (it runs on autograd's device thread, which cProfile does not see): wall time of
loss.backward() at a tiny batch (GPU time negligible) against the time spent inside this package's backward bodies.
    KIND=152 python tools/host_bwd_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
from convnets_amd import ops
KIND = int(os.environ.get('KIND', 152))
torch.manual_seed(21)
s = ca.Settings(kind=KIND, input_size=(3, 224, 224), num_classes=1000, batch_size=4, mixed_precision=True)
m = ca.MODELS['resnet'](s).cuda(); m.init_optimizer(); m.train()
x = torch.randn(4, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (4,), device='cuda')
acc = {}


def wrap(cls, name):
    orig = getattr(cls, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        key = cls.__name__ + '.' + name
        c, t = acc.get(key, (0, 0.0))
        acc[key] = (c + 1, t + time.perf_counter() - t0)
        return r
    setattr(cls, name, staticmethod(timed))


for cls in (ops.ConvBnActFn, ops.BatchNormActFn, ops.Conv2dFn, ops.BatchNormActMaxPoolFn, ops.LinearFn, ops.GlobalAvgPoolFn,
            ops.SoftmaxCrossEntropySumFn, ops.MaxPool2dFn):
    wrap(cls, 'backward')
lib = ops._lib.load()
for fn in ('cnk_conv2d_dgrad', 'cnk_conv2d_dgrad_bnfold', 'cnk_conv2d_wgrad', 'cnk_conv2d_wgrad_bnfold', 'cnk_bn_bwd', 'cnk_bn_bwd_coef',
           'cnk_conv2d_bnfold_prepare'):
    orig = getattr(lib, fn)

    def mk(orig, fn):
        def timed(*a):
            t0 = time.perf_counter()
            r = orig(*a)
            c, t = acc.get('lib.' + fn, (0, 0.0))
            acc['lib.' + fn] = (c + 1, t + time.perf_counter() - t0)
            return r
        return timed
    setattr(lib, fn, mk(orig, fn))
for _ in range(3):
    m.train_batch((x, y))
torch.cuda.synchronize()
acc.clear()
tb = tf = 0.0
N = 10
for _ in range(N):
    m.optimizer.zero_grad()
    t0 = time.perf_counter()
    loss, _ = m.training_step((x, y))
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    m.optimizer.step()
    tf += t1 - t0
    tb += t2 - t1
torch.cuda.synchronize()
print('forward %.2f ms, backward %.2f ms per step (host wall, batch 4)' % (tf / N * 1e3, tb / N * 1e3))
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('  %-40s %6.1f calls/step  %7.3f ms/step  %6.1f us/call' % (k, c / N, t / N * 1e3, t / c * 1e6))
