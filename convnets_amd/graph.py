"""hipGraph-captured training step (BASELINE config 5: "comm/compute overlap + hipGraph-captured step").

A ResNet-152 step is ~1,900 kernel launches issued from Python: 27 ms of host work against ~30 ms of GPU work at
bs 128, so the step is launch-bound as soon as anything else (a data loader, a gradient reducer) shares the host
thread.  `GraphedTrainStep` captures ONE iteration of the reference's loop body (src/basemodel.py:245-288 --
zero_grad, forward, CE(sum), backward, clipping, Adam) into a hipGraph and replays it with one launch per step.

What makes the step capturable:
  * no host sync inside the step: loss and score stay device scalars (basemodel.train_batch);
  * every buffer comes from the caller (the library never allocates), the workspaces are grow-only;
  * the two step-varying kernel arguments -- Adam's bias corrections and the dropout call counter -- are read from
    device memory that the host refreshes before each replay (ops.CaptureScalars, cnk_adam_step_dev,
    cnk_dropout_fwd_dev), so a replay computes exactly what the eager step would have computed;
  * the weight-gradient side stream forks from and re-joins the capturing stream through events, so it is part of
    the captured graph (two branches that the runtime may run concurrently, as in the eager step); the
    data-parallel reducer's stream and the RCCL all-reduces it issues (convnets_amd/dp.py) join the same way.
The learning rate is a kernel argument of the captured Adam launch; when the scheduler (src/basemodel.py:69-76) changes it the
replay does NOT re-capture (every capture after a process's first replays ~30 % slower) -- the kernel applies
lr / bc1 with bc1 read from the device record, so the record carries bc1 * lr_captured / lr_now instead.
"""
import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, example_batch, warmup=3, wgrad_share=None):
        """model: a MultiClassBaseModel on the GPU with its optimizer initialised; example_batch: (x, y) device
        tensors of the shapes every later batch will have.  The warm-up steps (lazy initialisation: packed
        weight tables, workspaces, allocator pools) run on a snapshot, so constructing the object does NOT
        advance training.
        wgrad_share: None keeps ops.WGRAD_SHARE; a float sets it; 'auto' MEASURES it (model.calibrate_wgrad_share) -- per
        candidate share of the weight gradients' split-K depth (ops.WGRAD_SHARE: how hard the side stream competes with
        the backward pass it runs beside) a few eager steps are timed on a snapshot, the fastest stays."""
        if model.optimizer is None:
            model.init_optimizer()
        x, y = example_batch
        if not (x.is_cuda and y.is_cuda):
            raise ValueError('GraphedTrainStep: the example batch must live on the GPU')
        self.model = model
        self.x = x.clone()
        self.y = y.clone()
        self.scalars = ops.CaptureScalars(x.device)
        self.graph = None
        self.lr = None
        self.replays = 0
        model.train()
        snap = self._snapshot()
        # The warm-up runs on the SAME side stream the capture uses: autograd remembers the stream a parameter's
        # AccumulateGrad node was created on and synchronises with it in every later backward -- a node left over from
        # the default stream would make the captured backward touch a stream outside the capture (a hard crash in
        # capture_end, not a Python error).
        self.stream = torch.cuda.Stream(device=x.device)
        self.stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                model.train_batch((self.x, self.y))
        torch.cuda.current_stream(x.device).wait_stream(self.stream)
        self._bn_layers = [m for m in model.modules() if hasattr(m, '_stats_epoch')]
        self._restore(snap)
        self.wgrad_share_ms = None
        if wgrad_share == 'auto':
            # measured on EAGER steps (same two-stream schedule), on the capture stream, before the one capture: capturing
            # the step several times in one process is not an option -- every capture after the first replayed 30 % slower
            # (26.7 vs 20.7 ms, ResNet-50: the runtime's stream-to-hardware-queue mapping)
            self.stream.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(self.stream):
                self.wgrad_share_ms = model.calibrate_wgrad_share((self.x, self.y)) or None
            torch.cuda.current_stream(x.device).wait_stream(self.stream)
        elif isinstance(wgrad_share, (int, float)):
            ops.WGRAD_SHARE[0] = float(wgrad_share)
        self._capture()

    @classmethod
    def build_collective(cls, model, example_batch, group=None, **kwargs):
        """The constructor as the COLLECTIVE operation it is under data parallelism: every rank of `group` calls this at the
        same point of its training loop (the first batch).  Warm-up steps, the split-K probe (its timings are all-reduced, so
        every rank picks the same share) and the captured step issue the same bucket all-reduces on every rank; afterwards
        the ranks agree on the outcome -- if the capture was refused on ANY rank, every rank drops its graph and trains
        eagerly (returns None), so no rank is left replaying collectives its peers issue eagerly in a different order.
        A rank that dies inside its warm-up leaves the others blocked in a collective, as with any data-parallel step."""
        import warnings
        import torch.distributed as dist
        step, err = None, None
        try:
            step = cls(model, example_batch, **kwargs)
        except (RuntimeError, NotImplementedError, ValueError) as e:      # a refused capture, not a dead process
            err = e
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if multi:
            dev = example_batch[0].device
            ok = torch.tensor([0 if step is None else 1], dtype=torch.int32,
                              device=dev if dist.get_backend(group) == 'nccl' else 'cpu')
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                step = None
        if step is None:
            warnings.warn('GraphedTrainStep: the hipGraph capture was refused%s; every rank trains with the eager step'
                          % ('' if err is None else ' on this rank (%s: %s)' % (type(err).__name__, err)))
        return step

    # ---- state that a throw-away step must not leave behind ----------------------------------------------
    def _snapshot(self):
        opt = self.model.optimizer
        return dict(p=opt.flat_p.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), step=opt.step_count,
                    bufs=[b.clone() for b in self.model.buffers()], drop=ops._dropout_calls[0])

    def _restore(self, snap):
        opt = self.model.optimizer
        with torch.no_grad():
            opt.flat_p.copy_(snap['p'])
            opt.exp_avg.copy_(snap['m'])
            opt.exp_avg_sq.copy_(snap['v'])
            for b, saved in zip(self.model.buffers(), snap['bufs']):
                b.copy_(saved)
        opt.step_count = snap['step']
        ops._dropout_calls[0] = snap['drop']
        self._mark_weights_changed()

    def _mark_weights_changed(self):
        ops.bump_weights_epoch()
        for m in self._bn_layers:
            m._stats_epoch += 1

    def _capture(self):
        opt = self.model.optimizer
        if len(opt.param_groups) != 1:
            raise NotImplementedError('GraphedTrainStep: one parameter group (the reference builds one, src/basemodel.py:64-67)')
        group = opt.param_groups[0]
        # The captured Adam launch steps by lr_captured / bc1 with bc1 read from the device record, which is how a later
        # learning-rate change reaches it (replay()).  A step captured at lr == 0 (warm-up schedules) would freeze that
        # quotient at 0 and turn it into 0 / 0 once the rate moves: capture such a step with a nominal rate of 1 instead;
        # replays at lr == 0 then carry bc1 = inf, i.e. a step of exactly zero, like the eager kernel.
        lr_now = group['lr']
        self.lr = lr_now if lr_now != 0 else 1.0
        # every other hyper-parameter is a kernel argument frozen into the graph: replay() refuses a change
        self._frozen = (tuple(group['betas']), group['eps'], group['weight_decay'])
        group['lr'] = self.lr
        try:
            self._capture_step(opt)
        finally:
            group['lr'] = lr_now

    def _capture_step(self, opt):
        step0, drop0 = opt.step_count, ops._dropout_calls[0]
        self.scalars.dropout_calls = 0
        self.graph = torch.cuda.CUDAGraph()
        ops.CAPTURE[0] = self.scalars
        # every eager collective / kernel issued so far has completed, and the capture checks only THIS thread:
        # the RCCL watchdog thread polls the events of eager collectives, a HIP call that a process-wide
        # ("global") capture would reject and abort the process on
        import gc
        self._drop_stale_autograd_state()
        # objects freed by that collection may have been registered convs of other models: bring the weight packer's
        # device table up to date BEFORE the capture (the upload is a host-to-device copy), and keep the collector off
        # while capturing so the registered set cannot change under it
        ops.packer_for(self.x.device).sync_table()
        # (a chained slab sum left pending by a backward pass that raised is performed HERE, eagerly -- inside the capture it
        # would be replayed with every step, into buffers that are released once it has run)
        ops.join_side_streams(self.x.device)
        torch.cuda.synchronize()
        self._drain_collective_watchdog()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode='thread_local'):
                self.loss, self.score = self.model.train_batch((self.x, self.y))
        finally:
            ops.CAPTURE[0] = None
            if gc_was_on:
                gc.enable()
        # capture ran the Python side of one step without executing it
        opt.step_count, ops._dropout_calls[0] = step0, drop0

    @staticmethod
    def _drain_collective_watchdog():
        """The process group's watchdog thread keeps every EAGER collective (the warm-up steps' bucket all-reduces) in a list
        and polls its completion event every 100 ms until it has seen it complete.  Once the capture's first all-reduce has
        pulled the communicator's stream into the capture, HIP refuses a query of an event that was last recorded on that
        stream -- although it was recorded long before the capture -- with hipErrorCapturedEvent, and the watchdog aborts the
        process (seen in ~40 % of the runs of test_graphed_dp_step_with_rccl_allreduce_equals_eager whenever the host reached
        that all-reduce within one polling interval of the device synchronisation above).  All those collectives HAVE
        completed (device synchronised): give the watchdog two polling intervals to drop them before the capture starts."""
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and 'nccl' in str(dist.get_backend()):
                import time
                time.sleep(0.25)
        except (RuntimeError, ValueError):
            pass

    def _drop_stale_autograd_state(self):
        """Nothing of an earlier eager step may survive into the capture: the loss module's references to its last
        outputs go, parameter gradients that are not views of the arena are detached, and a collection frees graphs
        kept alive only by cycles.  A caller that still HOLDS a non-detached loss / output of an eager step keeps that
        step's graph alive; its nodes are not replayed by this capture, so they are harmless as long as they are not
        run (backward) during it."""
        import gc
        cf = getattr(self.model, 'cost_function', None)
        for name in ('last_logits', 'last_pred'):
            if cf is not None and hasattr(cf, name):
                try:
                    setattr(cf, name, None)
                except Exception:
                    pass
        gc.collect()

    def replay(self, batch=None):
        """One training step on `batch` (None: the resident batch again).  Returns (loss, correct) as device
        scalars, the tensors of the captured step: read or accumulate them before the next replay."""
        opt = self.model.optimizer
        if batch is not None:
            x, y = batch
            self.x.copy_(x, non_blocking=True)
            self.y.copy_(y, non_blocking=True)
        bc = opt.bias_corrections(opt.step_count + 1)
        group = opt.param_groups[0]
        if len(opt.param_groups) != 1 or (tuple(group['betas']), group['eps'], group['weight_decay']) != self._frozen:
            raise RuntimeError('GraphedTrainStep: betas / eps / weight_decay / the parameter groups changed after the capture; '
                               'they are arguments of the captured Adam launch -- build a new GraphedTrainStep')
        lr = group['lr']
        if lr != self.lr:
            # adam_kernel steps by (lr_captured / bc1): a changed learning rate rides on the record (equal to the eager
            # step's lr / bc1 up to one rounding of the quotient)
            bc = (bc[0] * (self.lr / lr) if lr != 0 else float('inf'), bc[1])
        self.scalars.push(bc, ops._dropout_calls[0])
        self.graph.replay()
        opt.step_count += 1
        ops._dropout_calls[0] += self.scalars.dropout_calls
        self._mark_weights_changed()
        self.replays += 1
        return self.loss, self.score
