"""Does the weight-gradient branch of the captured step overlap the backward in a hipGraph REPLAY as it does in the eager step?
Times (untraced) per step: eager two-stream, eager one-stream, graph two-stream, graph one-stream, and the host duration of the
replay call itself.      python tools/probe_graph_overlap.py [KIND] [BS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
from convnets_amd import ops
KIND = int(sys.argv[1]) if len(sys.argv) > 1 else 50
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(21)
s = ca.Settings(kind=KIND, input_size=(3, 224, 224), num_classes=1000, batch_size=BS, mixed_precision=True)
m = ca.MODELS['resnet'](s).cuda(); m.init_optimizer(); m.train()
x = torch.randn(BS, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (BS,), device='cuda')


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(n):
        h = time.perf_counter(); fn(); host.append(time.perf_counter() - h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, sorted(host)[n // 2] * 1e3


for two in (True, False, True, False):
    ops.WGRAD_STREAM = two
    e = timed(lambda: m.train_batch((x, y)))
    g = ca.GraphedTrainStep(m, (x, y))
    r = timed(g.replay)
    print('wgrad stream %-5s  eager %.2f ms/step (host %.2f)   graph replay %.2f ms/step (host call %.2f ms)' % (two, e[0], e[1], r[0], r[1]), flush=True)
    del g
