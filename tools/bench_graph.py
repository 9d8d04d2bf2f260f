"""Eager step vs hipGraph-captured step (convnets_amd.GraphedTrainStep): ms/step and host issue time.
    python tools/bench_graph.py resnet 152 224 1000 128 bf16"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
family, kind, size, classes, bs, dt = sys.argv[1], (int(sys.argv[2]) if sys.argv[2].isdigit() else sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
torch.manual_seed(21)
s = ca.Settings(kind=kind, input_size=(3, size, size), num_classes=classes, batch_size=bs, mixed_precision=dt == 'bf16')
m = ca.MODELS[family](s).cuda(); m.init_optimizer(); m.train()
if 'RANK' in os.environ:            # python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/bench_graph.py ...
    import torch.distributed as dist
    from convnets_amd import dp
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    dp.attach(m)
x = torch.randn(bs, 3, size, size, device='cuda'); y = torch.randint(0, classes, (bs,), device='cuda')
n = int(os.environ.get('STEPS', 10))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        out = fn()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3, float(out[0])


e_ms, e_host, e_loss = timed(lambda: m.train_batch((x, y)))
g = ca.GraphedTrainStep(m, (x, y))
g_ms, g_host, g_loss = timed(lambda: g.replay())
print('%s-%s %dx%d bs%d %s: eager %.2f ms/step (host %.2f) loss %.4f | graph %.2f ms/step (host %.2f) loss %.4f | %.0f -> %.0f img/s' % (
    family, kind, size, size, bs, dt, e_ms, e_host, e_loss, g_ms, g_host, g_loss, bs / e_ms * 1e3, bs / g_ms * 1e3))
