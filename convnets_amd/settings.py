"""Minimal look-alike of the reference's Settings attribute bag (src/settings.py:169-319) carrying
the fields the hot path reads, with the reference's defaults (:177-222).  The reference's
hyper-parameter distributions, tuner hooks and reflection-based defaulting are out of scope
(SURVEY.md section 2, row 6).

Deviation (declared): `mixed_precision=True` selects bf16 operands with fp32
accumulation, fp32 BN statistics and fp32 master weights -- not the reference's CUDA fp16 autocast
+ dynamic GradScaler -- because BASELINE.json's ResNet configs ask for bf16 on MI355X.
"""
import torch


class DeviceMngr:
    """src/mngrdevice.py:14-43: first GPU if present else CPU; recursive non-blocking move."""

    def __init__(self, device=None):
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() \
                else torch.device('cpu')
        self.device = torch.device(device)

    def move(self, data):
        if isinstance(data, (list, tuple)):
            return [self.move(x) for x in data]
        return data.to(self.device, non_blocking=True)


class Settings:
    DEF = dict(batch_size=256, batch_norm=True, epochs=50, learning_rate=0.01, lr_factor=0.1,
               lr_patience=10, weight_decay=1e-4, dropout_rate=0.5, loss_optim=False,
               grad_clip_norm=False, gc_max_norm=1, grad_clip_value=False, gc_value=1,
               init_params=True, sanity_check=False, debug=False, mixed_precision=True, seed=21,
               test_sample_size=90,
               hipgraph=False,     # (not a reference field) train_model replays a captured hipGraph per batch shape
               # (not a reference field) the same under data parallelism (BASELINE config 5: "comm/compute overlap + hipGraph-captured
               # step"): the capture -- warm-up, split-K probe and the captured step all issue the bucket all-reduces -- is then a
               # COLLECTIVE operation entered by every rank at its first training batch (basemodel._graphed_batch).  None: the
               # environment decides (CNK_GRAPH_DP=1), default off: replays of captured RCCL nodes are verified on one rank only
               hipgraph_dp=None)

    def __init__(self, kind, input_size, num_classes, device=None, **kwargs):
        self.kind = kind
        self.input_size = tuple(input_size)
        self.num_classes = num_classes
        unknown = set(kwargs) - set(self.DEF)
        if unknown:
            raise TypeError('Settings: unknown field(s) %s' % sorted(unknown))
        for k, v in self.DEF.items():
            val = kwargs.get(k)
            setattr(self, k, v if val is None else val)     # None -> default, as settings.py:294-300
        self.device = device if isinstance(device, DeviceMngr) else DeviceMngr(device)
        if self.hipgraph_dp is None:
            import os
            self.hipgraph_dp = os.environ.get('CNK_GRAPH_DP', '0') == '1'

    @property
    def compute_dtype(self):
        return torch.bfloat16 if self.mixed_precision else torch.float32

    def to_dict(self):
        d = {k: getattr(self, k) for k in self.DEF}
        d.update(kind=self.kind, input_size=self.input_size, num_classes=self.num_classes)
        return d
