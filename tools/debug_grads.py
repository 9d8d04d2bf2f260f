"""Per-parameter gradient error of the HIP path vs the oracle (debug aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
from oracle import torch_ref as tr

kind, size, bs = 26, 32, 16
torch.manual_seed(21)
ref = tr.RefResNet(tr.RefSettings(kind, (3, size, size), 10, batch_size=bs, dropout_rate=0))
x, y = tr.synthetic_batch(bs, (3, size, size), 10, seed=21)
torch.manual_seed(21)
s = ca.Settings(kind=kind, input_size=(3, size, size), num_classes=10, batch_size=bs, dropout_rate=0, mixed_precision=False, device='cuda:0')
model = ca.ResNet(s).cuda()
model.init_optimizer(); model.train(); ref.train()
model.optimizer.zero_grad()
acts = {}
def hook(name):
    def f(m, i, o): acts[name] = o.detach().float().cpu()
    return f
for n, m in model.named_modules():
    if n in ('features.0.0', 'features.0', 'features.1', 'features.2'): m.register_forward_hook(hook(n))
racts = {}
for n, m in ref.named_modules():
    if n in ('features.0.0', 'features.0', 'features.1', 'features.2'): m.register_forward_hook((lambda name: (lambda mod, i, o: racts.__setitem__(name, o.detach().clone())))(n))
out = model(x.cuda()); loss = model.cost_function(out, y.cuda()); loss.backward()
oref = ref(x); lref = ref.cost_function(oref, y); lref.backward()
torch.cuda.synchronize()
for n in acts:
    print('act', n, float((acts[n] - racts[n]).norm() / racts[n].norm()))
errs = []
for (k, p), q in zip(model.named_parameters(), ref.parameters()):
    e = float((p.grad.cpu().double() - q.grad.double()).norm() / (q.grad.double().norm() + 1e-30))
    errs.append((e, k, float(q.grad.norm())))
print('loss', float(loss), float(lref))
for e, k, n in errs:
    print('%.3e  %-45s |g|=%.3e' % (e, k, n))
