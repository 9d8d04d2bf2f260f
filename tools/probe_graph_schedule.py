"""When do the weight gradients of the captured step run?  Device-clock stamps (cnk_debug_stamp, 100 MHz) on the compute stream
where each conv's incoming gradient is complete and on the weight-gradient stream around each weight-gradient launch, read after
an UNTRACED graph replay and after an eager step.  Prints, per conv in backward order, when its gradient was ready, when its
weight gradient started / ended, and the lag.      python tools/probe_graph_schedule.py [KIND] [BS]"""
import os, sys
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
for _ in range(3): m.train_batch((x, y))
torch.cuda.synchronize()
buf = torch.zeros(32768, dtype=torch.int64, device='cuda')


def report(title, log, first=0):
    t = buf.cpu().tolist()[first:first + len(log)]
    t0 = min(v for v in t if v)
    rows, cur = [], {}
    for lab, v in zip(log, t):
        cur[lab] = (v - t0) / 100.0          # us
        if lab == 'side_end':
            rows.append(cur); cur = {}
    print(title)
    print('  conv#   dy ready   wgrad start   wgrad end   start lag (us)')
    for i, r in enumerate(rows):
        if i < 6 or i % 6 == 0 or i >= len(rows) - 4:
            print('  %4d  %9.1f  %11.1f  %10.1f  %10.1f' % (i, r['main'], r['side_begin'], r['side_end'], r['side_begin'] - r['main']))
    print('  last dy ready %.1f us, last wgrad end %.1f us' % (rows[-1]['main'], rows[-1]['side_end']))


log = []
ops.STAMP[0] = (buf, log)
m.train_batch((x, y))
torch.cuda.synchronize()
report('eager step', log)
per_step = len(log)
del log[:]
buf.zero_()
g = ca.GraphedTrainStep(m, (x, y))          # warm-up steps, then the capture: its stamps are the LAST per_step entries, now graph nodes
n = len(log)
ops.STAMP[0] = None
buf.zero_()
for _ in range(3): g.replay()
torch.cuda.synchronize()
nz = (buf != 0).nonzero().flatten().cpu().tolist()
print('log entries %d (per step %d); slots written by the replays: %d .. %d (%d)' % (n, per_step, nz[0] if nz else -1, nz[-1] if nz else -1, len(nz)))
first = nz[0] if nz else 0
report('graph replay (untraced)', log[first:first + per_step], first)
