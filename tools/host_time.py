"""Host-side issue time of a training step vs its GPU time (is the step launch-bound?).
    python tools/host_time.py            # single process
    CNK_DP_FORCE=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
from convnets_amd import dp
dist_on = 'RANK' in os.environ
if dist_on:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get('CNK_DIST_BACKEND', 'nccl'), device_id=torch.device('cuda', 0))
torch.manual_seed(21)
KIND = int(os.environ.get('KIND', 50)); BS = int(os.environ.get('BS', 256))
s = ca.Settings(kind=KIND, input_size=(3, 224, 224), num_classes=1000, batch_size=BS, mixed_precision=True)
m = ca.MODELS['resnet'](s).cuda(); m.init_optimizer()
if dist_on or os.environ.get('CNK_DP_FORCE') == '1':
    dp.attach(m)
m.train()
x = torch.randn(BS, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (BS,), device='cuda')
for _ in range(3): m.train_batch((x, y))
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    h0 = time.perf_counter()
    m.train_batch((x, y))
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 10
# host-only cost: the same loop at a tiny batch (GPU work negligible)
print('step %.2f ms | host issue per step: median %.2f ms (min %.2f, max %.2f)' % (tot * 1e3, sorted(host)[5] * 1e3, min(host) * 1e3, max(host) * 1e3))
