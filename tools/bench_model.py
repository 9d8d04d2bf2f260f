"""Training-step timing of any model family of the package (tuning aid; bench.py is the headline benchmark).
    python tools/bench_model.py mobilenet 1 224 1000 256 bf16"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
family, kind, size, classes, bs, dt = sys.argv[1], (float(sys.argv[2]) if '.' in sys.argv[2] else int(sys.argv[2]) if sys.argv[2].isdigit() else sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
torch.manual_seed(21)
s = ca.Settings(kind=kind, input_size=(3, size, size), num_classes=classes, batch_size=bs, mixed_precision=dt == 'bf16')
m = ca.MODELS[family](s).cuda(); m.init_optimizer(); m.train()
x = torch.randn(bs, 3, size, size, device='cuda'); y = torch.randint(0, classes, (bs,), device='cuda')
for _ in range(3): m.train_batch((x, y))
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get('STEPS', 10))
for _ in range(n): m.train_batch((x, y))
torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / n
print('%s-%s %dx%d bs%d %s: %.2f ms/step, %.0f img/s' % (family, kind, size, size, bs, dt, dt_ * 1e3, bs / dt_))
