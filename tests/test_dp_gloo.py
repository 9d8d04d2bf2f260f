"""CPU suite: the N>1 data-parallel path (convnets_amd/dp.py) on world_size-2 gloo.

The reducer works on any flat gradient arena; here the arena is built on CPU around the oracle's
torch.nn ConvNet (no BN, so a data-parallel step is EXACTLY a single-process step on the
concatenated batch: CE(sum) => gradients add)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed):
    sys.path.insert(0, ROOT)
    from oracle import torch_ref as tr
    torch.manual_seed(seed)
    m = tr.RefConvNet(tr.RefSettings(0, (1, 28, 28), 10, batch_size=8, batch_norm=False, dropout_rate=0))
    params = list(m.parameters())
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 63) // 64 * 64
    flat_p, flat_g = torch.zeros(off), torch.zeros(off)
    for p, o in zip(params, offsets):
        flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
        p.data = flat_p[o:o + p.numel()].view(p.shape)
        p.grad = flat_g[o:o + p.numel()].view(p.shape)
    return m, params, offsets, flat_p, flat_g, tr


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    # different seeds per rank on purpose: the reducer must broadcast rank 0's replica
    m, params, offsets, flat_p, flat_g, tr = _build(seed=21 + rank)
    from convnets_amd.dp import GradReducer
    red = GradReducer(params, flat_g, offsets, bucket_bytes=64 << 10, flat_param=flat_p)
    assert len(red.buckets) >= 3 and red.buckets[0][1] == offsets[-1] + params[-1].numel()   # last layer first
    x, y = tr.synthetic_batch(16, (1, 28, 28), 10, seed=5)
    xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]
    m.train()
    for step in range(2):
        flat_g.zero_()
        red.begin_step()
        loss = m.cost_function(m(xs), ys)
        loss.backward()
        red.finish_step()
        with torch.no_grad():
            flat_p.add_(flat_g, alpha=-1e-3)          # plain SGD: identical on both ranks after the reduce
    dist.barrier()
    if rank == 0:
        torch.save({'flat_p': flat_p.clone(), 'flat_g': flat_g.clone()}, out)
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_global_batch(tmp_path):
    out = str(tmp_path / 'r0.pt')
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method='spawn')
    got = torch.load(out)
    m, params, offsets, flat_p, flat_g, tr = _build(seed=21)
    x, y = tr.synthetic_batch(16, (1, 28, 28), 10, seed=5)
    m.train()
    for step in range(2):
        flat_g.zero_()
        m.cost_function(m(x), y).backward()
        with torch.no_grad():
            flat_p.add_(flat_g, alpha=-1e-3)
    assert torch.allclose(got['flat_g'], flat_g, rtol=1e-4, atol=1e-5)
    assert torch.allclose(got['flat_p'], flat_p, rtol=1e-5, atol=1e-6)


def test_bucket_layout_covers_arena_once():
    sys.path.insert(0, ROOT)
    from convnets_amd.dp import GradReducer
    m, params, offsets, flat_p, flat_g, tr = _build(seed=1)
    red = GradReducer(params, flat_g, offsets, bucket_bytes=1 << 12)
    seen = sorted(i for _, _, ids in red.buckets for i in ids)
    assert seen == list(range(len(params)))
    for s, e, ids in red.buckets:
        assert s == min(offsets[i] for i in ids) and e == max(offsets[i] + params[i].numel() for i in ids)
    red.begin_step()
    m.cost_function(m(torch.randn(4, 1, 28, 28)), torch.tensor([1, 2, 3, 4])).backward()
    assert all(red._launched)                   # every bucket fired from the autograd hooks
    # a second report of the same parameter in one step (in-place kernels + torch's own hook) is ignored
    red.begin_step()
    red._param_ready(0)
    before = list(red._pending)
    red._param_ready(0)
    assert red._pending == before
    red.finish_step()
    red.remove()


def _collective_build_worker(rank, world, port, out):
    """GraphedTrainStep.build_collective on two ranks: a capture refused on ONE rank makes EVERY rank train eagerly (no rank
    is left replaying collectives its peer issues in another order); when both succeed both keep their graph."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import warnings
    from convnets_amd import graph
    results = []
    for fail_on in (1, None):
        def fake_init(self, model, batch, **kw):
            if fail_on is not None and dist.get_rank() == fail_on:
                raise RuntimeError('capture refused (test)')
            self.model = model
        orig = graph.GraphedTrainStep.__init__
        graph.GraphedTrainStep.__init__ = fake_init
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                step = graph.GraphedTrainStep.build_collective('model', (torch.zeros(1), torch.zeros(1)))
            results.append((step is not None, len(w)))
        finally:
            graph.GraphedTrainStep.__init__ = orig
    dist.barrier()
    torch.save(results, out % rank)
    dist.destroy_process_group()


def test_collective_graph_construction_agrees_across_ranks(tmp_path):
    out = str(tmp_path / 'c%d.pt')
    mp.start_processes(_collective_build_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method='spawn')
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0[0] == (False, 1) and r1[0] == (False, 1), 'rank 1 refused: both ranks must drop the graph (and say so)'
    assert r0[1] == (True, 0) and r1[1] == (True, 0)
