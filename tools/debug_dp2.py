import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import convnets_amd as ca
from convnets_amd import dp, ops
from oracle import torch_ref as tr
os.environ['CNK_DP_FORCE'] = '0'
torch.manual_seed(21)
s = ca.Settings(kind=18, input_size=(3, 32, 32), num_classes=10, batch_size=8, dropout_rate=0, mixed_precision=False, learning_rate=1e-3)
m = ca.ResNet(s).cuda(); m.init_optimizer(); m.train()
red = dp.attach(m, bucket_bytes=256 << 10)
names = [k for k, _ in m.named_parameters()]
events = []
orig_ready = red._param_ready
def traced(i):
    events.append(('ready', names[i] if i is not None else None))
    orig_ready(i)
red._param_ready = traced
orig_launch = red._launch
def l2(b):
    if not red._launched[b]:
        events.append(('LAUNCH', b, [names[i] for i in red.buckets[b][2]]))
    orig_launch(b)
red._launch = l2
# hooks were registered with the original bound method via lambda -> they call self._param_ready (instance attr lookup) fine
x, y = tr.synthetic_batch(8, (3, 32, 32), 10, seed=5)
m.train_batch((x, y))
torch.cuda.synchronize()
cnt = collections.Counter(e[1] for e in events if e[0] == 'ready')
print('params notified != 1:', {k: v for k, v in cnt.items() if v != 1}, 'missing:', [n for n in names if n not in cnt])
for e in events:
    if e[0] == 'LAUNCH':
        seen = set(ev[1] for ev in events[:events.index(e)] if ev[0] == 'ready')
        miss = [n for n in e[2] if n not in seen]
        print('LAUNCH bucket', e[1], 'n=%d' % len(e[2]), 'MISSING at launch:' if miss else 'ok', miss[:4])
