import os, socket, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(rank, world, port, out, bucket, sync):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import convnets_amd as ca
    from convnets_amd import dp
    from oracle import torch_ref as tr
    torch.manual_seed(21)
    s = ca.Settings(kind=18, input_size=(3, 32, 32), num_classes=10, batch_size=8, dropout_rate=0, mixed_precision=False, learning_rate=1e-3)
    m = ca.ResNet(s).cuda(); m.init_optimizer(); m.train()
    red = dp.attach(m, bucket_bytes=bucket)
    if sync:
        orig = red._launch
        def l2(b):
            torch.cuda.synchronize(); orig(b); torch.cuda.synchronize()
        red._launch = l2
    x, y = tr.synthetic_batch(16, (3, 32, 32), 10, seed=5)
    loss, _ = m.train_batch((x[rank*8:(rank+1)*8], y[rank*8:(rank+1)*8]))
    torch.cuda.synchronize()
    names = [k for k, _ in m.named_parameters()]
    torch.save(dict(g=m.optimizer.flat_g.cpu(), offs=m.optimizer.offsets, names=names, numels=[p.numel() for p in m.parameters()], nb=len(red.buckets)), out % rank)
    dist.barrier(); dist.destroy_process_group()

if __name__ == '__main__':
    import torch.multiprocessing as mp
    for bucket, sync in [(256 << 10, 0), (256 << 10, 1), (1 << 30, 0)]:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        out = '/tmp/dbg_r%d.pt'
        mp.start_processes(worker, args=(2, port, out, bucket, sync), nprocs=2, join=True, start_method='spawn')
        a, b = torch.load(out % 0), torch.load(out % 1)
        bad = []
        for name, o, n in zip(a['names'], a['offs'], a['numels']):
            d = (a['g'][o:o+n] - b['g'][o:o+n]).abs()
            if float(d.max()) > 0: bad.append((name, int((d > 0).sum()), n, float(d.max()), float(a['g'][o:o+n].abs().max())))
        print('bucket=%d sync=%d buckets=%d: %d/%d params differ' % (bucket, sync, a['nb'], len(bad), len(a['names'])))
        for r in bad[:6] + bad[-3:]: print('   ', r)
