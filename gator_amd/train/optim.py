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
        ops.Deferred.flush(o.flat.device)                  # the queued weight gradients: one grouped GEMM launch
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
        dev = self.flat.device                             # (a backward that raised may have left entries of THIS device behind)
        ops.Deferred.queue[:] = [e for e in ops.Deferred.queue if e[2].device != dev]
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
        """MultiStepLR as main/train.py drives it: `lr_scheduler.step()` after every epoch, epochs counted from begin_epoch = 1
        (lib/core/config.py:71, main/train.py:36-39), so epoch e runs with gamma^(milestones <= e - 1): lr drops from epoch 31."""
        done = max(0, self.epoch - 1)
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if done >= m)

    def step(self, grad):
        self.step_count += 1
        f = self.p.flat
        st = ctypes.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
        _lib.check(_lib.load().gator_t_adam(f.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.p.numel,
                                            float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count,
                                            self.device_step.data_ptr() if self.device_step is not None else None, st), 'gator_t_adam')

    # ---- checkpoint interchange with the reference (main/train.py:51-58 saves optimizer.state_dict(); base.py:73-77 loads it) ----
    def state_dict(self):
        """torch.optim.Adam.state_dict() layout: per-parameter {step, exp_avg, exp_avg_sq} in model.parameters() order."""
        state = {}
        for i, (a, b, shape) in enumerate(self.p.slots):
            state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.exp_avg[a:b].view(shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[a:b].view(shape).clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'initial_lr': self.base_lr,
                 'params': list(range(len(self.p.slots)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        if len(sd['param_groups']) != 1 or len(sd['param_groups'][0]['params']) != len(self.p.slots):
            raise ValueError('optimizer state does not match the parameter list (%d tensors expected)' % len(self.p.slots))
        g = sd['param_groups'][0]
        self.base_lr = float(g.get('initial_lr', g['lr']))
        self.betas, self.eps = tuple(g['betas']), float(g['eps'])
        steps = set()
        for i, (a, b, shape) in enumerate(self.p.slots):
            st = sd['state'].get(i)
            if st is None:
                continue
            if tuple(st['exp_avg'].shape) != tuple(shape):
                raise ValueError('optimizer state %d has shape %s, parameter %s has %s' % (i, tuple(st['exp_avg'].shape), self.p.names[i], shape))
            dev = self.exp_avg.device
            ops.raw_unary(ops.U_AFFINE, st['exp_avg'].to(dev).float(), 1.0, 0.0, out=self.exp_avg[a:b].view(shape))
            ops.raw_unary(ops.U_AFFINE, st['exp_avg_sq'].to(dev).float(), 1.0, 0.0, out=self.exp_avg_sq[a:b].view(shape))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ: %s' % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        if self.device_step is not None:
            self.device_step.fill_(self.step_count)
