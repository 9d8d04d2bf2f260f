"""Inference throughput with the reference's own test() procedure (src/basemodel.py:601-722: eval mode,
50 warm-up forwards, per-batch device-event timing of forward + arg-max), on the configs its
data/results.xlsx reports (32x32 inputs, 10 classes, batch 256) plus ResNet-50 at 224x224.
Prints thousands of images per second, the unit of the reference's fps column."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnets_amd as ca

PUBLISHED = {'vgg11': 18.369, 'resnet26': 18.835}      # data/results.xlsx rows 2 and 4 (unnamed CUDA GPU, fp16 AMP)
CASES = [('vgg11', 'vgg', 11, (3, 32, 32), 10, 256), ('resnet26', 'resnet', 26, (3, 32, 32), 10, 256),
         ('resnet50-224', 'resnet', 50, (3, 224, 224), 1000, 256)]
out = []
for name, family, kind, in_size, classes, bs in CASES:
    torch.manual_seed(21)
    s = ca.Settings(kind=kind, input_size=in_size, num_classes=classes, batch_size=bs, mixed_precision=True)
    model = ca.MODELS[family](s).cuda()
    nb = 40 if in_size[1] == 32 else 8
    g = torch.Generator().manual_seed(1)

    class Loader(list):
        dataset = range(nb * bs)
    loader = Loader([(torch.randn(bs, *in_size, generator=g), torch.randint(0, classes, (bs,), generator=g))
                     for _ in range(nb)])
    scores, times, fps = model.test(loader, repetitions=50)
    out.append({'model': name, 'batch': bs, 'dtype': 'bf16', 'k_images_per_sec': round(fps, 3),
                'ms_per_batch_median': round(sorted(times)[len(times) // 2], 3),
                'published_k_images_per_sec_other_hw': PUBLISHED.get(name)})
    print(json.dumps(out[-1]), flush=True)
