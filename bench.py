"""Headline benchmark: images/sec of one full training step (zero_grad + forward + CE(sum) loss +
backward [+ gradient all-reduce] + Adam step -- the reference's loop body, src/basemodel.py:245-288)
for ResNet-50 224x224, bs=256 per GPU, bf16 operands / fp32 accumulate, synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events
recorded on the launch stream around the dominant conv kernel class (cnk_prof_*), `cpu_baseline`
is the oracle's torch-CPU restatement of the same step timed on this box's host cores (rank 0,
N=1 only, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# The step uses four HIP streams (compute, weight gradients, gradient all-reduce, RCCL's own); the runtime
# multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share a queue
# serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}     # MI355X_MICROARCH.md: dense MFMA peaks
CONV_CLASSES = ['conv3x3_fwd', 'conv3x3_dgrad', 'conv3x3_wgrad', 'conv1x1_fwd', 'conv1x1_dgrad',
                'conv1x1_wgrad']
# kernel family -> the cnk_prof classes its launches fall in (HIP events around every launch of the class).
# 3x3 stride-1 convs run in igemm_halo3x3_kernel, every other igemm conv (1x1, strided 3x3) in
# igemm_gather_kernel; both serve the forward AND the data gradient.  The dominant family of the step
# (largest summed launch time) is the one the roofline object describes, priced against the roofline that
# binds it: MFMA if flops/peak_flops > bytes/peak_bytes for its launches, else HBM.
# kernel families timed per LAUNCH (libcnk's kernel-level scopes) -> what each one runs in a training step
KERNEL_NOTE = {'igemm_gather_kernel': '1x1 conv fwd+dgrad, strided 3x3 convs',
               'igemm_halo3x3_kernel': 'stride-1 3x3 conv fwd+dgrad (plain and weight-prefetching variant)',
               'wgrad_mfma_kernel': 'conv weight gradients, 3x3 and 1x1',
               'bn_apply_kernel': 'BN training forward apply (+ReLU, +residual)',
               'bn_bwd_apply_kernel': 'BN backward apply'}
PEAK_HBM_GBPS = 8000.0                           # MI355X_MICROARCH.md: HBM3E spec bandwidth
MODEL_CFG = {
    'resnet50': ('resnet', 50, (3, 224, 224), 1000, 256),
    'resnet152': ('resnet', 152, (3, 224, 224), 1000, 128),
    'vgg16': ('vgg', 16, (3, 224, 224), 1000, 64),
    'alexnet': ('alexnet', 0, (3, 224, 224), 1000, 128),
    'resnet26s': ('resnet', 26, (3, 32, 32), 10, 64),         # quick plumbing check
}


def prof_collect(lib, cls_bit):
    n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    rc = lib.cnk_prof_collect(cls_bit, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
    assert rc == 0
    return n.value, ms.value, fl.value, by.value


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM);
    bench.py cannot drive the profiler itself, so this is the last profiled value, or None."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None
    try:
        tab = json.load(open(path))
        for k, v in tab.get('families', {}).items():
            if k and k in kernel:
                return dict(v, source='profiles/pmc_traffic.json (%s)' % tab.get('tag', '?'))
    except Exception:
        return None
    return None


def cpu_baseline(model_key, dtype_note):
    """The oracle ('port' of the reference's torch-CPU step) timed on the host cores."""
    from oracle import torch_ref as tr
    family, kind, in_size, classes, _bs_default = MODEL_CFG[model_key]
    nproc = os.cpu_count() or 1
    try:
        nproc = len(os.sched_getaffinity(0))             # the cores this process may be scheduled on
    except (AttributeError, OSError):
        pass
    # ... of which a container may only be granted a share (cgroup v2 cpu.max / v1 cfs quota): more threads than that
    # share only fight each other (a first run with one thread per visible core did not finish a step in 7 minutes)
    quota = None
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: None if t.split()[0] == 'max' else float(t.split()[0]) / float(t.split()[1])),
                        ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            quota = parse(open(path).read().strip())
            break
        except Exception:
            continue
    granted = nproc if not quota else max(1, min(nproc, int(quota + 0.5)))
    threads = granted                                    # all host cores this process is granted
    torch.set_num_threads(threads)
    print('bench: CPU baseline on %d threads (nproc %d, cgroup cpu share %s) ...' % (threads, nproc, quota), file=sys.stderr, flush=True)
    # the metric's batch (bs 256) if it fits and stays inside the bounded sample, else bs 64 (throughput is flat in
    # batch on CPU: 13-18 img/s either way on the 16 granted cores).  One warm-up step decides: it must neither run out of
    # memory nor take longer than WARM_LIMIT seconds (two more timed steps of that length have to fit the bench's few minutes).
    WARM_LIMIT = 30.0
    fallback = None
    for bs in ((_bs_default, 64) if _bs_default != 64 else (64,)):
        try:
            torch.manual_seed(21)
            m = tr.MODELS[family](tr.RefSettings(kind, in_size, classes, batch_size=bs, dropout_rate=0.5))
            m.init_optimizer()
            x, y = tr.synthetic_batch(bs, in_size, classes, seed=21)
            t0 = time.perf_counter()
            tr.ref_training_step(m, x, y)                      # warm-up
            warm = time.perf_counter() - t0
        except (MemoryError, RuntimeError) as e:               # torch reports a failed host allocation as RuntimeError
            if bs == 64:
                raise
            fallback = 'bs %d did not fit (%s)' % (bs, type(e).__name__)
            m = x = y = None
            import gc
            gc.collect()
            continue
        print('bench: CPU warm-up step at bs %d: %.2f s' % (bs, warm), file=sys.stderr, flush=True)
        if bs != 64 and warm > WARM_LIMIT:
            fallback = 'bs %d warm-up step took %.0f s (> %.0f s): outside the bounded sample' % (bs, warm, WARM_LIMIT)
            m = x = y = None
            continue
        break
    times = []
    t_end = time.perf_counter() + 20.0
    nmin, nmax = (2, 3) if bs != 64 else (3, 5)
    while len(times) < nmin or (time.perf_counter() < t_end and len(times) < nmax):
        t0 = time.perf_counter()
        tr.ref_training_step(m, x, y)
        times.append(time.perf_counter() - t0)
        print('bench: CPU step %d: %.2f s' % (len(times), times[-1]), file=sys.stderr, flush=True)
    times.sort()
    med = times[(len(times) - 1) // 2]
    cfg = torch.__config__.show()
    build = '; '.join(l.strip(' -') for l in cfg.splitlines() if any(k in l for k in ('oneAPI', 'oneDNN', 'MKL-DNN', 'OpenMP', 'CPU capability')))
    return {'value': round(bs / med, 3), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'nproc': nproc, 'cgroup_cpus': quota, 'threads': torch.get_num_threads(), 'torch': torch.__version__, 'torch_build': build[:400],
            'batch': bs, 'batch_fallback': fallback,
            'sample': '%d fp32 training steps of %s at bs=%d on torch-CPU (oracle/torch_ref.py) after one warm-up step, median'
                      % (len(times), model_key, bs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='resnet50', choices=sorted(MODEL_CFG))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: BASELINE config)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', action='store_true', help='print the per-class kernel time table to stderr')
    ap.add_argument('--graph', choices=['auto', 'on', 'off'], default='auto',
                    help='time hipGraph replays of the step (convnets_amd.GraphedTrainStep: one launch per step) or the '
                    'eager step.  auto = on for a single process; under torch.distributed on only with CNK_GRAPH_DP=1 (the '
                    'replay of captured RCCL all-reduces is verified on one rank only; the capture is then a collective '
                    'operation, GraphedTrainStep.build_collective).  Per-kernel HIP events cannot be read inside a replay, '
                    'so with the graph the roofline launch durations come from eager steps after the timed region')
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank if torch.cuda.device_count() > local_rank else 0)
        backend = os.environ.get('CNK_DIST_BACKEND', 'nccl')      # 'nccl' IS RCCL on ROCm; gloo only for 1-GPU rehearsals
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        distributed = True
    else:
        distributed = False
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    import convnets_amd as ca
    from convnets_amd import _lib, dp
    lib = _lib.load()

    family, kind, in_size, classes, def_bs = MODEL_CFG[args.model]
    bs = args.batch or def_bs
    torch.manual_seed(21)                                    # identical replicas
    setting = ca.Settings(kind=kind, input_size=in_size, num_classes=classes, batch_size=bs,
                          mixed_precision=args.dtype == 'bf16', device=dev)
    model = ca.MODELS[family](setting).to(dev)
    model.init_optimizer()
    if world > 1 or os.environ.get('CNK_DP_FORCE') == '1':
        dp.attach(model)
    model.train()
    g = torch.Generator(device=dev).manual_seed(21 + rank)  # each rank its own shard of synthetic data
    x = torch.randn(bs, *in_size, generator=g, device=dev)
    y = torch.randint(0, classes, (bs,), generator=g, device=dev)
    batch = (x, y)

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; the last warm-up step doubles as the per-class breakdown that picks the dominant
    # conv kernel class for the roofline object
    # (that breakdown step runs the weight gradients on the compute stream, so every kernel's duration is
    # its own; the timed region below uses the two-stream backward, where durations overlap)
    from convnets_amd import ops as _ops
    for i in range(max(args.warmup, 1)):
        last = i == max(args.warmup, 1) - 1
        if last:
            torch.cuda.synchronize()
            overlap, _ops.WGRAD_STREAM = _ops.WGRAD_STREAM, False
            lib.cnk_prof_begin(0x1FF | sum(_lib.KF.values()))
        loss, _ = model.train_batch(batch)
        if last:
            torch.cuda.synchronize()
            table = {name: prof_collect(lib, _lib.CLS[name]) for name in _lib.CLS}
            ktable = {name: prof_collect(lib, bit) for name, bit in _lib.KF.items()}
            lib.cnk_prof_end()
            _ops.WGRAD_STREAM = overlap
    dom_kernel = max(ktable, key=lambda k: ktable[k][1])        # the kernel with the largest share of the step
    dom_mask = _lib.KF[dom_kernel]
    if args.breakdown and rank == 0:
        for name, (n, ms, fl, by) in sorted(list(table.items()) + list(ktable.items()), key=lambda kv: -kv[1][1]):
            print('  %-22s n=%4d  %8.3f ms  %8.1f TFLOP/s  %8.1f GB/s(alg)' %
                  (name, n, ms, fl / ms / 1e9 if ms else 0, by / ms / 1e6 if ms else 0), file=sys.stderr)

    graphed = None
    if args.graph == 'on' or (args.graph == 'auto' and (not distributed or os.environ.get('CNK_GRAPH_DP') == '1')):
        # (under torch.distributed every rank enters here together: warm-up, split-K probe and capture issue the bucket
        # all-reduces, and the ranks agree on the outcome -- a refused capture anywhere means the eager step everywhere)
        graphed = ca.GraphedTrainStep.build_collective(model, batch, wgrad_share='auto')
    share_ms = graphed.wgrad_share_ms if graphed is not None else None
    if graphed is None and args.dtype == 'bf16':
        # eager step (data parallel runs): the same measurement on eager steps, every rank alike
        share_ms = model.calibrate_wgrad_share(batch) or None
    if graphed is not None:
        # one captured hipGraph launch per step; per-kernel events cannot be read inside a replay, so the
        # dominant kernel's launch durations are measured on eager steps AFTER the timed region
        # the W warm-up steps above ran eagerly (capture needs them); the clocks and the caches then sit through the capture,
        # so the replays get W untimed steps of their own (first replays after a capture run ~2 % slow)
        for _ in range(max(2, args.warmup)):
            graphed.replay()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = graphed.replay()
        sync()
        elapsed = time.perf_counter() - t0
        lib.cnk_prof_begin(dom_mask)
        for _ in range(min(args.steps, 5)):
            model.train_batch(batch)
        torch.cuda.synchronize()
    else:
        lib.cnk_prof_begin(dom_mask)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = model.train_batch(batch)
        sync()
        elapsed = time.perf_counter() - t0
    n_l, ms_l, fl_l, by_l = prof_collect(lib, dom_mask)
    lib.cnk_prof_end()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss)
    assert loss_val == loss_val, 'loss is NaN'

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * bs * args.steps / elapsed
        tflops = fl_l / ms_l / 1e9 if ms_l > 0 else 0.0            # algorithmic flops / launch time
        gbps = by_l / ms_l / 1e6 if ms_l > 0 else 0.0               # algorithmic bytes / launch time
        peak_tf = PEAK_TFLOPS[args.dtype]
        if tflops / peak_tf >= gbps / PEAK_HBM_GBPS:
            bound, achieved, peak, unit = 'mfma', tflops, peak_tf, 'TFLOP/s'
        else:
            bound, achieved, peak, unit = 'hbm', gbps, PEAK_HBM_GBPS, 'GB/s'
        # every kernel family against the roofline that binds it (durations: HIP events recorded on the launch stream
        # around each launch, serial breakdown step = weight gradients on the compute stream, eager)
        by_kernel = []
        for kname, (n_k, ms_k, fl_k, by_k) in sorted(ktable.items(), key=lambda kv: -kv[1][1]):
            if not n_k or ms_k <= 0:
                continue
            tf_k, gb_k = fl_k / ms_k / 1e9, by_k / ms_k / 1e6
            mf = tf_k / peak_tf >= gb_k / PEAK_HBM_GBPS
            by_kernel.append({'kernel': kname, 'kernel_runs': KERNEL_NOTE[kname], 'bound': 'mfma' if mf else 'hbm',
                              'achieved': round(tf_k if mf else gb_k, 2), 'peak': peak_tf if mf else PEAK_HBM_GBPS,
                              'unit': 'TFLOP/s' if mf else 'GB/s', 'frac': round((tf_k / peak_tf) if mf else (gb_k / PEAK_HBM_GBPS), 4),
                              'launches_per_step': n_k, 'avg_launch_us': round(ms_k / n_k * 1e3, 2), 'step_ms': round(ms_k, 3),
                              'duration_source': 'HIP events on the launch stream around every launch (cnk_prof_*), serial eager step'})
        # the same family in the committed rocprofv3 passes (tools/run_profiles.sh -> profiles/pmc_traffic.json): its average
        # launch duration there (all launches of a profiled run, mostly inside graph replays) next to the live HIP-event one
        traffic = pmc_traffic(dom_kernel)
        rocprof_cross = None
        if traffic and traffic.get('avg_launch_us') and n_l:
            alg = by_l / n_l if bound == 'hbm' else fl_l / n_l
            ach = alg / (traffic['avg_launch_us'] * 1e-6) / (1e9 if bound == 'hbm' else 1e12)
            rocprof_cross = {'avg_launch_us': traffic['avg_launch_us'], 'achieved': round(ach, 2), 'frac': round(ach / peak, 4),
                             'traffic_over_algorithmic': round(traffic['total'] / (by_l / n_l), 3) if by_l else None,
                             'note': 'algorithmic bytes (or flops) per launch of THIS run / the family\'s average launch duration in the '
                                     'committed rocprofv3 --kernel-trace --stats summary; same launch population as `traffic_detail`'}
        # whole step against both roofs: algorithmic conv + head flops of the step (the C-ABI class scopes of
        # the serial breakdown step) over the measured step time, and the step's HBM-side bytes from the committed PMC passes
        step_flops = sum(table[c][2] for c in table)
        step_tf = step_flops / (ms_per_step * 1e-3) / 1e12
        step_traffic = None
        try:
            step_traffic = (json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get('step') or {}).get('bytes')
        except Exception:
            pass
        step_level = {'step_tflops': round(step_tf, 1), 'step_tflops_frac': round(step_tf / peak_tf, 4),
                      'step_algorithmic_tflop': round(step_flops / 1e12, 3),
                      'step_hbm_gbytes': round(step_traffic / 1e9, 2) if step_traffic and args.model == 'resnet50' and bs == def_bs else None,
                      'step_hbm_frac': round(step_traffic / (ms_per_step * 1e-3) / (PEAK_HBM_GBPS * 1e9), 4)
                      if step_traffic and args.model == 'resnet50' and bs == def_bs else None,
                      'note': 'flops: algorithmic, all conv / linear classes of one step; bytes: profiles/pmc_traffic.json `step` (sum over '
                              'kernels of PMC bytes per launch x launches per step, headline config only) / this run\'s step time'}
        conv_classes = {c: {'tflops': round(table[c][2] / table[c][1] / 1e9, 1), 'frac_of_mfma_peak': round(table[c][2] / table[c][1] / 1e9 / peak_tf, 4),
                            'calls_per_step': table[c][0], 'step_ms': round(table[c][1], 3)} for c in CONV_CLASSES if table[c][1] > 0}
        out = {
            'metric': 'images/sec training step, ResNet-50 224x224 bs=256/GPU' if args.model == 'resnet50'
                      else 'images/sec training step, %s bs=%d/GPU' % (args.model, bs),
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%s %dx%d training step (fwd+CE(sum)+bwd+Adam), bs=%d/GPU, random-init'
                                   % (args.model, in_size[1], in_size[2], bs),
                       'global_batch': world * bs, 'parallelism': 'dp%d' % world,
                       'final_loss': round(loss_val, 4)},
            'roofline': {'bound': bound, 'kernel': dom_kernel, 'kernel_runs': KERNEL_NOTE[dom_kernel], 'achieved': round(achieved, 2), 'peak': peak,
                         'unit': unit, 'frac': round(achieved / peak, 4),
                         # `traffic`: HBM bytes per launch (FETCH_SIZE corrected + WRITE_SIZE, separate --pmc passes), a number
                         # as the bench contract words it; the passes' breakdown beside it
                         'traffic': traffic.get('total') if traffic else None, 'traffic_detail': traffic, 'rocprof': rocprof_cross,
                         'launches': n_l, 'avg_launch_ms': round(ms_l / n_l, 4) if n_l else None,
                         'avg_launch_gflop': round(fl_l / n_l / 1e9, 3) if n_l else None,
                         'avg_launch_alg_mbytes': round(by_l / n_l / 1e6, 2) if n_l else None,
                         'two_stream_backward': bool(_ops.WGRAD_STREAM and args.dtype == 'bf16'),
                         'wgrad_share': {'chosen': _ops.WGRAD_SHARE[0], 'ms_per_step_by_share': share_ms and {str(k): round(v, 3) for k, v in share_ms.items()},
                                         'note': 'split-K depth of the side-stream weight gradients, measured at start-up (ops.WGRAD_SHARE)'},
                         'hipgraph_step': graphed is not None,
                         'launch_durations_from': 'eager steps after the timed region' if graphed is not None else 'the timed region',
                         'serial_step_ms_by_kernel': {k: round(ktable[k][1], 3) for k in ktable},
                         'serial_launches_by_kernel': {k: ktable[k][0] for k in ktable},
                         'serial_conv_tflops_by_class': {c: round(table[c][2] / table[c][1] / 1e9, 1) for c in CONV_CLASSES if table[c][1] > 0},
                         'duration_source': 'HIP events recorded on the launch stream around every launch of the kernel (libcnk cnk_prof_*); '
                                            + ('eager two-stream steps after the timed hipGraph replays' if graphed is not None else 'the timed eager steps')
                                            + '; cross-check: profiles/ rocprofv3 --kernel-trace --stats of the same command'},
            # every timed kernel family against its own roofline, and the conv classes (C-ABI call scopes: the 3x3 MFMA
            # fraction north_star names is conv3x3_fwd / conv3x3_dgrad here)
            'roofline_by_kernel': by_kernel,
            'conv_classes_serial': conv_classes,
            'step_level': step_level,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.model, args.dtype)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
