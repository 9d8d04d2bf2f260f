"""cProfile of the host side of the eager training step at a tiny batch (GPU time negligible): where the Python /
ctypes issue time of a step goes.      KIND=50 python tools/host_profile.py"""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca
KIND = int(os.environ.get('KIND', 50))
torch.manual_seed(21)
s = ca.Settings(kind=KIND, input_size=(3, 224, 224), num_classes=1000, batch_size=4, mixed_precision=True)
m = ca.MODELS['resnet'](s).cuda(); m.init_optimizer(); m.train()
x = torch.randn(4, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (4,), device='cuda')
for _ in range(3):
    m.train_batch((x, y))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    m.train_batch((x, y))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumtime').print_stats(22)
