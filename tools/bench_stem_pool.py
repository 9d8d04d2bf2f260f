import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convnets_amd import _lib
if os.environ.get('CNK_LIB_OVERRIDE'):
    _lib.LIB_PATH = os.environ['CNK_LIB_OVERRIDE']
lib = _lib.load()
N, H, W, C = 256, 112, 112, 64
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
xs = [torch.randn(N, H, W, C, device='cuda').bfloat16() for _ in range(3)]
sc, sf = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.3
y = torch.empty(N, 56, 56, C, dtype=torch.bfloat16, device='cuda'); idx = torch.empty(N, 56, 56, C, dtype=torch.uint8, device='cuda')
def run(i):
    assert lib.cnk_maxpool2d_fwd_bn(P(xs[i % 3]), P(sc), P(sf), 1, P(y), P(idx), N, H, W, C, 3, 3, 2, 2, 1, 1, _lib.CNK_BF16, st) == 0
for i in range(5): run(i)
torch.cuda.synchronize()
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10): run(i)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
print('stem pool forward %.1f us (%s)' % (statistics.median(ts), ' '.join('%.0f' % t for t in ts)), 'y', float(y.float().sum()), 'idx', int(idx.long().sum()))
