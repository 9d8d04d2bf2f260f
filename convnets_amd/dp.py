"""Data-parallel gradient reduction for the training step: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; north_star adds data parallelism.  The
design follows from the reference's loss: CrossEntropyLoss(reduction='sum') (src/basemodel.py:46)
makes a step on the concatenated global batch equal to the SUM (not the mean) of per-rank
gradients, so the collective is all-reduce(SUM).  BN statistics stay rank-local, like stock DDP.

Gradients live in one flat arena (convnets_amd/optim.py); a bucket is a contiguous slice of it,
cut in REVERSE parameter order because backward produces the last layers' gradients first.  When
the last gradient of a bucket has been accumulated (post-accumulate-grad hook, fired on the
autograd thread), the bucket's all-reduce is issued on a side stream that waits on an event
recorded on the compute stream -- so RCCL traffic over xGMI overlaps the rest of backward -- and
the compute stream waits for all buckets only before the optimizer step.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU): a ring all-reduce is bound by one link, so buckets are large
(default 32 MiB, ~4 for ResNet-50's 102 MB of fp32 gradients) rather than NVSwitch-sized.
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, flat_grad, offsets, bucket_bytes=32 << 20, process_group=None,
                 flat_param=None):
        self.params = list(params)
        self.flat_grad = flat_grad
        self.offsets = list(offsets)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cuda = flat_grad.is_cuda
        import os
        self.force = os.environ.get('CNK_DP_FORCE') == '1'      # run the collective even on 1 rank (path test)
        if self.cuda:
            from . import ops as _o
            self.side = _o.comm_stream(flat_grad.device)       # (not torch.cuda.Stream(): pool streams repeat, ops.wgrad_stream)
        else:
            self.side = None
        # buckets from the END of the arena (last layers first)
        self.buckets = []            # (start_elem, end_elem, [param indices])
        idxs, nbytes, end = [], 0, None
        for i in reversed(range(len(self.params))):
            n = self.params[i].numel()
            if end is None:
                end = self.offsets[i] + n
            idxs.append(i)
            nbytes += 4 * n
            if nbytes >= bucket_bytes or i == 0:
                self.buckets.append((self.offsets[i], end, idxs))
                idxs, nbytes, end = [], 0, None
        self.bucket_of = {}
        for b, (_, _, ids) in enumerate(self.buckets):
            for i in ids:
                self.bucket_of[i] = b
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                       for i, p in enumerate(self.params) if p.requires_grad]
        # kernels that accumulate straight into the arena (convnets_amd.ops) report here instead
        from . import ops
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._listener = lambda p: self._param_ready(self._index.get(id(p)))
        ops._grad_ready_listeners.append(self._listener)
        self._pending, self._launched, self._works, self._seen = None, None, None, None
        self._compute_stream = None
        if flat_param is not None and self.world > 1:
            dist.broadcast(flat_param, src=0, group=process_group)      # identical replicas
            # the arena was overwritten behind autograd: packed weight images / folded inference images built from
            # the pre-broadcast weights (any forward before attach) must not be reused
            from . import ops as _ops
            _ops.bump_weights_epoch()

    def _param_ready(self, i):
        """Idempotent per parameter and step: a gradient written in place by a kernel is reported
        by convnets_amd.ops, and torch (2.10) ALSO fires the post-accumulate hook for a parameter
        whose Function returned None -- only the first report counts."""
        if self._pending is None or i is None or self._seen[i]:
            return
        self._seen[i] = True
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _make_hook(self, i):
        from . import ops

        def hook(param):
            if not ops.grad_sum_pending(param):          # (a chained slab sum: ops reports the gradient once it is issued)
                self._param_ready(i)
        return hook

    def begin_step(self):
        self._compute_stream = torch.cuda.current_stream(self.flat_grad.device) if self.cuda else None
        self._pending = [sum(1 for i in ids if self.params[i].requires_grad) for _, _, ids in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._seen = [False] * len(self.params)
        self._works = []

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        if self.world == 1 and not self.force:
            return
        s, e, _ = self.buckets[b]
        view = self.flat_grad[s:e]
        if self.cuda:
            from . import ops
            dev = self.flat_grad.device
            # the bucket's gradients were written on the compute stream and on the wgrad stream
            streams = {st.cuda_stream: st for st in [torch.cuda.current_stream(dev), self._compute_stream] + ops.side_streams(dev)
                       if st is not None}
            for st in streams.values():
                if ops.CAPTURE[0] is not None and not ops._is_capturing(st):
                    continue          # hipGraph capture: that stream carries none of the captured step's work
                ev = torch.cuda.Event()
                ev.record(st)
                self.side.wait_event(ev)
            if ops.CAPTURE[0] is not None and not ops._is_capturing(self.side):
                # (the collective would run eagerly beside the capture, on a communicator stream that IS capturing: the
                # process group's watchdog then aborts the process on the captured event -- fail here, with a traceback)
                raise RuntimeError('GradReducer: a bucket became ready during a hipGraph capture on a stream outside the capture '
                                   '(current %s, compute %s)' % (torch.cuda.current_stream(dev), self._compute_stream))
            with torch.cuda.stream(self.side):
                w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, None))
        else:
            self._works.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True), None))

    def finish_step(self):
        """Issue any bucket whose gradients never all arrived, then make the compute stream wait
        for every collective (no host block on CUDA: Work.wait() is a stream dependency)."""
        for b in range(len(self.buckets)):
            self._launch(b)
        for w, _ in self._works:
            w.wait()
        if self.cuda and self._works:
            torch.cuda.current_stream().wait_stream(self.side)
        self._pending = None

    def remove(self):
        from . import ops
        for h in self._hooks:
            h.remove()
        if self._listener in ops._grad_ready_listeners:
            ops._grad_ready_listeners.remove(self._listener)


def attach(model, bucket_bytes=32 << 20, process_group=None):
    """Wires a GradReducer to a model whose init_optimizer() has built the flat arena."""
    opt = model.optimizer
    import os
    if os.environ.get('CNK_DP_BUCKET_MB'):
        bucket_bytes = int(float(os.environ['CNK_DP_BUCKET_MB']) * (1 << 20))
    model.grad_reducer = GradReducer(opt.params, opt.flat_g, opt.offsets, bucket_bytes,
                                     process_group, flat_param=opt.flat_p)
    if model.grad_reducer.world > 1:
        # attaching mid-training is well defined: rank 0's Adam moments, step count and buffers (BN running
        # statistics) too; from here on the running statistics evolve rank-locally (no SyncBN in the reference)
        for t in (opt.exp_avg, opt.exp_avg_sq):
            dist.broadcast(t, src=0, group=process_group)
        step = torch.tensor([opt.step_count], dtype=torch.int64, device=opt.flat_p.device)
        dist.broadcast(step, src=0, group=process_group)
        opt.step_count = int(step.item())
        for b in model.buffers():
            dist.broadcast(b, src=0, group=process_group)
        for m in model.modules():
            if hasattr(m, '_stats_epoch'):
                m._stats_epoch += 1
    return model.grad_reducer
