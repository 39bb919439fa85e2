"""Parameters in ONE flat device buffer + torch.optim.Adam's update as one kernel launch (lib/funcs_utils.py:91-95: Adam, lr
only; MultiStepLR of :102-103).  The flat layout is also the data-parallel bucket: one all-reduce of `grad` per step."""
import ctypes

import numpy as np
import torch

from .. import _lib
from . import ops
from .model import is_buffer


class _Unflatten(torch.autograd.Function):
    """flat [n] -> one view per parameter; the backward writes every gradient into ONE flat gradient buffer (library copies)."""

    @staticmethod
    def forward(ctx, flat, owner):
        ctx.owner = owner
        ctx.set_materialize_grads(False)
        return tuple(flat[a:b].view(shape) for (a, b, shape) in owner.slots)

    @staticmethod
    def backward(ctx, *gs):
        o = ctx.owner
        out = o.gflat
        for g, (a, b, shape) in zip(gs, o.slots):
            if g is not None and g.data_ptr() != out.data_ptr() + 4 * a:      # ops that know their slot wrote it in place (ops.grad_slot)
                ops.raw_unary(ops.U_AFFINE, g, 1.0, 0.0, out=out[a:b].view(shape))
        return out, None


class FlatParams:
    def __init__(self, state_dict, device):
        """state_dict: reference-layout name -> array/tensor.  Parameters go into `flat`; buffers stay separate tensors."""
        self.names, self.slots, self.buffers = [], [], {}
        chunks, off = [], 0
        for k in state_dict:                                   # insertion order = the reference's state_dict order
            v = torch.as_tensor(np.asarray(state_dict[k])) if not torch.is_tensor(state_dict[k]) else state_dict[k]
            if is_buffer(k):
                self.buffers[k] = v.to(device).float().contiguous() if v.is_floating_point() else v.to(device)
                continue
            n = v.numel()
            self.names.append(k)
            self.slots.append((off, off + n, tuple(v.shape)))
            chunks.append(v.reshape(-1).float())
            off += (n + 3) // 4 * 4                            # 16-byte aligned slots
        self.numel = off
        flat = torch.zeros(off, dtype=torch.float32)
        for c, (a, b, _) in zip(chunks, self.slots):
            flat[a:b] = c
        self.flat = flat.to(device).requires_grad_(True)

    def views(self):
        """name -> differentiable view of the flat buffer (call once per step, inside the graph).  Every view carries its slice
        of this step's flat gradient buffer (`_gslot`): linear / LayerNorm backward write weight gradients straight into it."""
        self.gflat = ops.zeros((self.numel,), self.flat.device)
        vs = _Unflatten.apply(self.flat, self)
        for v, (a, b, shape) in zip(vs, self.slots):
            v._gslot = self.gflat[a:b].view(shape)
        return dict(zip(self.names, vs))

    def state_dict(self):
        sd = {k: self.flat.detach()[a:b].view(shape).clone() for k, (a, b, shape) in zip(self.names, self.slots)}
        sd.update({k: v.clone() for k, v in self.buffers.items()})
        return sd


class Adam:
    """torch.optim.Adam(params, lr) (betas 0.9 / 0.999, eps 1e-8, no weight decay) with MultiStepLR(milestones, gamma)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, milestones=(30,), gamma=0.1):
        self.p = params
        self.base_lr, self.betas, self.eps, self.milestones, self.gamma = lr, betas, eps, tuple(milestones), gamma
        self.exp_avg = ops.zeros((params.numel,), params.flat.device)
        self.exp_avg_sq = ops.zeros((params.numel,), params.flat.device)
        self.step_count, self.epoch = 0, 0
        self.device_step = None                      # device uint64 step counter (hipGraph replay): overrides step_count in the kernel

    @property
    def lr(self):
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if self.epoch >= m)

    def step(self, grad):
        self.step_count += 1
        f = self.p.flat
        st = ctypes.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
        _lib.check(_lib.load().gator_t_adam(f.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.p.numel,
                                            float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count,
                                            self.device_step.data_ptr() if self.device_step is not None else None, st), 'gator_t_adam')
