"""Adam over one flat fp32 arena (cnk_adam_step), the optimizer the reference configures at
src/basemodel.py:64-67: torch.optim.Adam(lr, weight_decay) -- L2 decay added to the gradient,
betas (0.9, 0.999), eps 1e-8.

MI355X-first layout: every parameter is a view into ONE contiguous HBM buffer, every .grad a view
into a second one.  That makes (a) the optimizer a single streaming kernel instead of 161 small
ones for ResNet-50, (b) zero_grad one memset, (c) data-parallel gradient buckets contiguous
slices that RCCL can reduce in place (convnets_amd/dp.py).
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatAdam:
    def __init__(self, params, lr=0.01, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('FlatAdam: no parameters')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on the MI355X HIP path only; move the model to the GPU '
                               'before init_optimizer() (reference order: mngrdevice.move, then fit)')
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError('FlatAdam: all parameters must be fp32 on one device')
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += _align(p.numel())
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
        self.param_groups = [dict(params=self.params, lr=lr, weight_decay=weight_decay, betas=betas,
                                  eps=eps)]
        self.step_count = 0
        ops.bump_weights_epoch()

    def grad_view(self, i):
        o, n = self.offsets[i], self.params[i].numel()
        return self.flat_g[o:o + n]

    def zero_grad(self, set_to_none=False):
        # grads stay views of the arena (never None): autograd then accumulates in place
        ops.join_side_streams(self.flat_g.device)
        self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self):
        ops.join_side_streams(self.flat_g.device)       # weight gradients written on the wgrad stream
        g = self.param_groups[0]
        self.step_count += 1
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        arenas = (ctypes.c_void_p(self.flat_p.data_ptr()), ctypes.c_void_p(self.flat_g.data_ptr()),
                  ctypes.c_void_p(self.exp_avg.data_ptr()), ctypes.c_void_p(self.exp_avg_sq.data_ptr()))
        cap = ops.CAPTURE[0]
        if cap is None:
            check(_lib.load().cnk_adam_step(*arenas, self.numel, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                                            g['weight_decay'], self.step_count, s))
        else:       # hipGraph capture: the bias corrections of the replayed step come from device memory
            check(_lib.load().cnk_adam_step_dev(*arenas, self.numel, g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                                                g['weight_decay'], cap.bias_ptr, s))
        ops.bump_weights_epoch()

    def bias_corrections(self, step):
        """(1 - beta1^step, sqrt(1 - beta2^step)) exactly as cnk_adam_step computes them."""
        out = (ctypes.c_float * 2)()
        g = self.param_groups[0]
        _lib.load().cnk_adam_bias_corrections(g['betas'][0], g['betas'][1], int(step), out)
        return float(out[0]), float(out[1])

    @torch.no_grad()
    def clip_grad_norm(self, max_norm):
        """clip_grad_norm_(parameters, max_norm) of basemodel.py:156 over the whole arena in two
        kernels; returns the total norm as a DEVICE scalar (no host sync)."""
        lib = _lib.load()
        ops.join_side_streams(self.flat_g.device)
        ws = ops.workspace(lib.cnk_grad_clip_workspace_bytes(), self.flat_g.device)
        norm = torch.empty((), dtype=torch.float32, device=self.flat_g.device)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.cnk_grad_clip_norm(ctypes.c_void_p(self.flat_g.data_ptr()), self.numel, float(max_norm),
                                     ctypes.c_void_p(norm.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                     ws.numel(), s))
        return norm

    @torch.no_grad()
    def clip_grad_value(self, clip_value):
        """clip_grad_value_(parameters, clip_value) of basemodel.py:171 as one kernel."""
        ops.join_side_streams(self.flat_g.device)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(_lib.load().cnk_grad_clip_value(ctypes.c_void_p(self.flat_g.data_ptr()), self.numel,
                                              float(clip_value), s))

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(),
                    exp_avg_sq=self.exp_avg_sq.clone(),
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != 'params'}])

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups[0].update(sd['param_groups'][0])
