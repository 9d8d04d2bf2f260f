"""Host-side mirror of the reference's MultiClassBaseModel for the hot path (src/basemodel.py).

Same method names, keyword arguments and error behaviour as the reference's layer factory
(:966-1042), shape tracker (:725-816), init_params (:85-108) and training-step sequence
(:180-215, :231-346), so model definitions written against the reference read the same here.
What differs is what the factory RETURNS: subclasses of the stock modules whose forward runs the
hand-written HIP kernels (convnets_amd/layers.py).  Out of scope: fit()'s
epoch driver, checkpoints, sklearn reports, plotting, tuner.
"""
import numpy as np
import torch
import torch.nn as nn

from . import layers, ops
from .optim import FlatAdam


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class MultiClassBaseModel(nn.Module):
    def __init__(self, setting):
        super().__init__()
        self.setting = setting
        self.model_name = self.__class__.__name__ + str(setting.kind)
        self.in_channels, self.height, self.width = setting.input_size      # basemodel.py:41-43
        self.cost_function = layers.CrossEntropyLoss(reduction='sum')       # basemodel.py:46
        self.optimizer = None
        self.grad_reducer = None          # data-parallel hook (convnets_amd/dp.py); None on 1 GPU

    # ---- optimizer / init (basemodel.py:58-108) ------------------------------------------------
    def init_optimizer(self):
        """Adam(lr, weight_decay) as basemodel.py:64-67, over a flat HBM arena.  The reference's
        ReduceLROnPlateau / GradScaler belong to fit() (out of scope; bf16 needs no loss scaling)."""
        self.optimizer = FlatAdam(self.parameters(), lr=self.setting.learning_rate,
                                  weight_decay=self.setting.weight_decay)
        # all BN step counters as views of one int64 arena: `num_batches_tracked += 1` for every
        # layer becomes one kernel per step (flushed in train_batch / before state_dict())
        bns = [m for m in self.modules() if isinstance(m, layers.BatchNorm2d)]
        if bns:
            arena = torch.stack([m.num_batches_tracked.detach().to(self.optimizer.flat_p.device) for m in bns])
            self._bn_counter_arena, self._bn_pending = arena, []
            for i, m in enumerate(bns):
                m.num_batches_tracked = arena[i]
                m._deferred_counter = [0]
                self._bn_pending.append(m._deferred_counter)
        return self.optimizer

    def flush_bn_counters(self):
        """Adds each BN layer's pending training-forward count to its num_batches_tracked: one
        kernel for the whole model (a scalar add in the usual case of equal counts)."""
        pend = getattr(self, '_bn_pending', None)
        if not pend:
            return
        counts = [c[0] for c in pend]
        if any(counts):
            if min(counts) == max(counts):
                self._bn_counter_arena += counts[0]
            else:
                self._bn_counter_arena += torch.tensor(counts, dtype=self._bn_counter_arena.dtype,
                                                       device=self._bn_counter_arena.device)
            for c in pend:
                c[0] = 0

    def state_dict(self, *args, **kwargs):
        self.flush_bn_counters()
        return super().state_dict(*args, **kwargs)

    def init_params(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        ops.bump_weights_epoch()
        return out

    # ---- step pieces (basemodel.py:111-229) -----------------------------------------------------
    @torch.no_grad()
    def score(self, outputs, targets):
        """Count of correct predictions (basemodel.py:119-121) -- kept on the device; the caller
        decides when to pay the host sync."""
        cf = self.cost_function
        ref = getattr(cf, 'last_logits', None)
        if ref is not None and ref() is outputs and cf.last_pred is not None:
            return (cf.last_pred == targets).sum()          # argmax came with the loss kernel
        return (outputs.argmax(dim=1) == targets).sum()

    def gradient_clipping(self):
        """basemodel.py:131-178: norm clipping, then value clipping, each over ALL gradients -- one
        pass over the flat gradient arena instead of one launch per parameter.  Returns the total
        norm (device scalar) when norm clipping is on."""
        if self.optimizer is None:
            raise RuntimeError('gradient_clipping: call init_optimizer() first (the gradients live in its arena)')
        total_norm = None
        if self.setting.grad_clip_norm:
            total_norm = self.optimizer.clip_grad_norm(self.setting.gc_max_norm)
        if self.setting.grad_clip_value:
            self.optimizer.clip_grad_value(self.setting.gc_value)
        return total_norm

    def forward_step(self, batch):
        inputs, targets = self.setting.device.move(batch)
        outputs = self(inputs)
        loss = self.cost_function(outputs, targets)
        return loss, self.score(outputs, targets)

    @torch.enable_grad()
    def training_step(self, batch):
        return self.forward_step(batch)

    @torch.no_grad()
    def validation_step(self, batch):
        return self.forward_step(batch)

    def train_batch(self, batch):
        """One iteration of the reference's loop body (basemodel.py:245-288): zero_grad, forward,
        CE(sum) loss, backward, [gradient all-reduce], clipping (off by default), optimizer step.
        Returns (loss, correct) as DEVICE scalars."""
        self.optimizer.zero_grad()
        if self.grad_reducer is not None:
            self.grad_reducer.begin_step()
        loss, score = self.training_step(batch)
        loss.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.finish_step()
        self.gradient_clipping()
        self.optimizer.step()
        self.flush_bn_counters()
        return loss.detach(), score

    def calibrate_wgrad_share(self, batch, candidates=(1.0, 0.75, 0.5), steps=3, reps=3):
        """Measures ops.WGRAD_SHARE for the EAGER step of this model (see convnets_amd/ops.py; GraphedTrainStep does the same
        for the captured step): per candidate `reps` timings of `steps` steps on a snapshot of the training state, the median
        counts.  Under data parallelism every rank must call it (the steps issue the gradient all-reduces).  Returns {share: ms}.
        The share is the fp32 summation order of the split-K weight gradients, i.e. it decides the bits of a run: the choice is
        kept on the model (`self.wgrad_share`), logged, and CNK_WGRAD_SHARE in the environment PINS it (no measurement) for
        runs that must reproduce bit for bit across boxes."""
        import logging
        import os
        import time
        from . import ops
        from .graph import GraphedTrainStep
        if self.optimizer is None:
            self.init_optimizer()
        if not (ops.WGRAD_STREAM and self.setting.mixed_precision):
            return {}
        if os.environ.get('CNK_WGRAD_SHARE'):
            ops.WGRAD_SHARE[0] = self.wgrad_share = float(os.environ['CNK_WGRAD_SHARE'])
            logging.getLogger('convnets_amd').info('wgrad split-K share pinned by CNK_WGRAD_SHARE=%s', ops.WGRAD_SHARE[0])
            return {}
        probe = GraphedTrainStep.__new__(GraphedTrainStep)          # borrow its snapshot / restore of the training state
        probe.model = self
        probe._bn_layers = [m for m in self.modules() if hasattr(m, '_stats_epoch')]
        self.train()
        snap = probe._snapshot()
        batch = self.setting.device.move(batch)
        timings = {}
        previous = ops.WGRAD_SHARE[0]
        try:
            for share in candidates:
                ops.WGRAD_SHARE[0] = share
                self.train_batch(batch)
                samples = []
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        self.train_batch(batch)
                    torch.cuda.synchronize()
                    samples.append((time.perf_counter() - t0) / steps * 1e3)
                samples.sort()
                timings[share] = samples[len(samples) // 2]
                probe._restore(snap)
        except BaseException:
            # a failed probe step must leave neither a candidate share nor a half-trained state behind
            ops.WGRAD_SHARE[0] = previous
            probe._restore(snap)
            raise
        red = getattr(self, 'grad_reducer', None)
        if red is not None and getattr(red, 'world', 1) > 1:
            # every rank must take the SAME share: the split-K depth is the fp32 summation order of the weight gradients, and
            # replicas that sum in different orders drift apart bit by bit.  Slowest rank per candidate decides.
            import torch.distributed as dist
            t = torch.tensor([timings[c] for c in candidates], dtype=torch.float64, device=self.setting.device.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=getattr(red, 'group', None))
            timings = {c: float(v) for c, v in zip(candidates, t.tolist())}
        # The SMALLEST share within 1 % of the fastest candidate.  The probe times eager steps; where those are host-bound (ResNet-152:
        # ~1,900 launches per step) every candidate measures the same and the choice used to fall to 1.0 by its tie-break -- the
        # worst one for the captured step (25.6-26.1 ms at 0.5, 26.3-26.4 at 1.0, A/B in one call).  Where the share matters to
        # the eager step it shows by far more than 1 % (VGG-16: -10 % at 0.5), and the captured ResNet-50 step is as fast at 0.5
        # as at 0.75.
        fastest = min(timings.values())
        best = min(c for c in timings if timings[c] <= 1.01 * fastest)
        ops.WGRAD_SHARE[0] = self.wgrad_share = best
        logging.getLogger('convnets_amd').info('wgrad split-K share %s chosen from %s (ms per eager step; CNK_WGRAD_SHARE pins it)',
                                               best, {c: round(v, 3) for c, v in timings.items()})
        return timings

    @torch.enable_grad()
    def train_model(self, dataloader):
        """basemodel.py:231-302 with the two per-step host syncs (:121, :287) replaced by
        device-side accumulation and ONE sync per epoch; returned values are identical."""
        self.train()
        if self.optimizer is None:
            self.init_optimizer()
        epoch_loss = torch.zeros((), dtype=torch.float64, device=self.setting.device.device)
        epoch_score = torch.zeros((), dtype=torch.int64, device=self.setting.device.device)
        for batch in dataloader:
            loss, score = self._graphed_batch(batch) if self.setting.hipgraph else self.train_batch(batch)
            epoch_loss += loss.double()
            epoch_score += score
            if self.setting.sanity_check:
                break
        n = len(dataloader.dataset)
        return epoch_loss.item() / n, epoch_score.item() / n

    def _graphed_batch(self, batch):
        """setting.hipgraph: the loop body as one hipGraph launch (convnets_amd/graph.py), one captured graph per
        batch shape (the last, smaller batch of an epoch gets its own).
        Under data parallelism (setting.hipgraph_dp) constructing the graph is a COLLECTIVE operation -- its warm-up steps, the
        split-K probe and the captured step itself issue the bucket all-reduces -- so there is exactly one point where it
        happens: the first training batch of the model, which every rank reaches together.  No rank can enter a capture
        alone afterwards: a batch of another shape runs the EAGER step (it issues the same collectives in the same order as
        a replay, so ranks may even mix the two), and a learning-rate change rides on the device record (graph.py)."""
        from .graph import GraphedTrainStep
        red = getattr(self, 'grad_reducer', None)
        dp = red is not None and (red.world > 1 or red.force)
        if dp and red.world > 1 and not self.setting.hipgraph_dp:
            return self.train_batch(batch)        # multi-rank replays of captured RCCL nodes: opt-in (Settings.hipgraph_dp)
        x, y = self.setting.device.move(batch)
        steps = self.__dict__.setdefault('_graphed_steps', {})
        key = (tuple(x.shape), x.dtype, tuple(y.shape))
        if dp:
            if not self.__dict__.get('_graph_dp_built'):
                self.__dict__['_graph_dp_built'] = True
                steps[key] = GraphedTrainStep.build_collective(self, (x, y), group=red.group if red.world > 1 else None)
            st = steps.get(key)
            return st.replay((x, y)) if st is not None else self.train_batch((x, y))
        if key not in steps:
            steps[key] = GraphedTrainStep(self, (x, y))
        return steps[key].replay((x, y))

    @torch.no_grad()
    def valid_model(self, dataloader):
        """basemodel.py:304-346."""
        self.eval()
        epoch_loss = torch.zeros((), dtype=torch.float64, device=self.setting.device.device)
        epoch_score = torch.zeros((), dtype=torch.int64, device=self.setting.device.device)
        for batch in dataloader:
            loss, score = self.validation_step(batch)
            epoch_loss += loss.double()
            epoch_score += score
            if self.setting.sanity_check:
                break
        n = len(dataloader.dataset)
        return epoch_loss.item() / n, epoch_score.item() / n

    # ---- inference measurement (basemodel.py:579-722) ---------------------------------------------
    def inference_time(self, total_times, num_instances):
        """basemodel.py:579-599, same arithmetic: `total_times` are per-batch device times in
        MILLISECONDS (cuda.Event.elapsed_time), so the returned `images_per_sec` = instances / sum of
        milliseconds -- i.e. THOUSANDS of images per second, the unit of data/results.xlsx's fps
        column."""
        total_times = np.array(total_times)
        entire_time = np.sum(total_times)
        mean_time = np.mean(total_times / self.setting.batch_size)
        std_time = np.std(total_times / self.setting.batch_size)
        return entire_time, mean_time, std_time, num_instances / entire_time

    @torch.no_grad()
    def test(self, dataloader, repetitions=50):
        """The measurement core of the reference's test() (basemodel.py:601-722): eval mode,
        `repetitions` warm-up forwards on a random batch, then every batch timed with device events
        around forward + arg-max.  Returns (scores, total_times, fps) like the reference: accuracy per
        subset of test_sample_size parts, per-batch milliseconds, and inference_time()'s fps.  The
        sklearn classification report / confusion matrix are out of scope."""
        self.eval()
        dev = self.setting.device
        bs, (ch, h, w) = self.setting.batch_size, self.setting.input_size
        rand_input = dev.move(torch.randn(bs, ch, h, w, dtype=torch.float))
        for _ in range(repetitions):
            self(rand_input)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        total_times, preds, targs = [], [], []
        for batch in dataloader:
            inputs, targets = dev.move(batch)
            start.record()
            outputs = self(inputs)
            predictions = torch.max(outputs, dim=1)[1]
            end.record()
            torch.cuda.synchronize()
            total_times.append(start.elapsed_time(end))
            preds.append(predictions.cpu().numpy())
            targs.append(targets.cpu().numpy())
            if self.setting.sanity_check:
                break
        y_preds, y_targets = np.concatenate(preds), np.concatenate(targs)
        n = len(dataloader.dataset)
        parts = getattr(self.setting, 'test_sample_size', None) or 1
        part = max(int(n / parts), 1)
        scores = [float(np.mean(y_targets[i:i + part] == y_preds[i:i + part]))
                  for i in range(0, len(y_targets), part)]
        _, _, _, fps = self.inference_time(total_times, n)
        return scores, total_times, fps

    # ---- shape tracker (basemodel.py:725-816) ---------------------------------------------------
    def save_conv_outshape(self, layer):
        kh, kw = _pair(layer.kernel_size)
        sh, sw = _pair(layer.stride)
        ph, pw = _pair(layer.padding)
        dh, dw = _pair(layer.dilation) if hasattr(layer, 'dilation') else (1, 1)
        self.height = int(np.floor(1 + (self.height + 2 * ph - dh * (kh - 1) - 1) / sh))
        self.width = int(np.floor(1 + (self.width + 2 * pw - dw * (kw - 1) - 1) / sw))
        if isinstance(layer, nn.Conv2d):
            self.in_channels = layer.out_channels
        return self.in_channels, self.height, self.width

    def save_adapt_outshape(self, layer):
        self.height, self.width = _pair(layer.output_size)
        return self.in_channels, self.height, self.width

    def num_flat_features(self):
        return self.in_channels * self.height * self.width

    def get_learning_rate(self):
        return self.optimizer.param_groups[0]['lr']

    def update_learning_rate(self):
        self.optimizer.param_groups[0]['lr'] = self.setting.learning_rate

    # ---- layer factory: THE drop-in boundary (basemodel.py:966-1042) ----------------------------
    def conv2d(self, in_channels=None, num_filters=None, set_output=True, **kwargs):
        if in_channels is None:
            in_channels = self.in_channels
        layer = layers.Conv2d(in_channels, num_filters, **kwargs, bias=not self.setting.batch_norm)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def conv2d_depthwise(self, in_channels=None, num_filters=None, set_output=True, **kwargs):
        """basemodel.py:981-997: nn.Conv2d(in, filters, groups=in); filters == in (every caller in the reference)
        runs on the depthwise kernels, a channel multiplier on the dense grouped path."""
        if in_channels is None:
            in_channels = self.in_channels
        if num_filters is None:
            num_filters = self.in_channels
        layer = layers.Conv2d(in_channels, num_filters, groups=in_channels, **kwargs,
                              bias=not self.setting.batch_norm)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def conv2d_block(self, in_channels=None, num_filters=None, set_output=True, activation=True,
                     **kwargs):
        mods = [self.conv2d(in_channels, num_filters, set_output, **kwargs)]
        if self.setting.batch_norm:
            mods += [layers.BatchNorm2d(num_features=num_filters)]
        if activation:
            mods += [layers.ReLU(inplace=True)]
        return layers.ConvBlock(*mods)

    def maxpool2d(self, set_output=True, **kwargs):
        layer = layers.MaxPool2d(**kwargs)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def avgpool2d(self, set_output=True, **kwargs):           # basemodel.py:1026-1033
        layer = layers.AvgPool2d(**kwargs)
        if set_output:
            self.save_conv_outshape(layer)
        return layer

    def adapt_avgpool2d(self, set_output=True, **kwargs):
        layer = layers.AdaptiveAvgPool2d(**kwargs)
        if set_output:
            self.save_adapt_outshape(layer)
        return layer

    # ---- forward: features -> flatten -> classifier (resnet.py:110-117 et al.) -------------------
    def forward(self, x):
        if x.dim() != 4 or tuple(x.shape[1:]) != tuple(self.setting.input_size):
            raise ValueError('expected input [N,%d,%d,%d], got %s' % (*self.setting.input_size, tuple(x.shape)))
        if not (x.dtype == self.setting.compute_dtype and x.is_contiguous(memory_format=torch.channels_last)):
            x = ops.nchw_to_nhwc(x, self.setting.compute_dtype)     # NCHW fp32 batch -> NHWC compute dtype
        # (else: already prepared on the device by ops.input_pipeline)
        if not self.__dict__.get('_pools_linked'):
            layers.link_block_pools(self)                       # conv -> BN -> ReLU -> MaxPool2d: BN inside the pooling pass
            self.__dict__['_pools_linked'] = True
        x = self.features(x)
        x = torch.flatten(x, start_dim=1).float()               # head runs in fp32
        return self.classifier(x)
