"""The five training losses of the reference (lib/core/loss.py:10-118, combined at lib/core/base.py:137-148) on the device.
Each loss kernel produces the value AND the gradient w.r.t. the prediction in one pass (include/gator_train.h); the autograd
node only scales that stored gradient.  Per-vertex gradients of the face losses are gathered in a fixed order (no atomics)."""
import ctypes

import numpy as np
import torch

from .. import _lib
from . import ops

_I64x4 = ctypes.c_int64 * 4


def vertex_incidence(faces, num_verts):
    """CSR lists vertex -> (3*face + corner), ascending: the gather order of the face-loss gradients."""
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    flat = f.reshape(-1)
    order = np.argsort(flat, kind='stable')
    ptr = np.zeros(num_verts + 1, np.int64)
    np.add.at(ptr, flat + 1, 1)
    return np.cumsum(ptr).astype(np.int32), order.astype(np.int32)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, kind, weight, owner, target, valid):
        if not pred.is_cuda:
            raise RuntimeError('gator_amd.train.losses: predictions must live on a HIP device (there is no CPU path)')
        p = pred if pred.is_contiguous() else ops.raw_copy(pred)
        # the kernels read target / valid as device float32: anything else (a host tensor, a float64 / bool mask from a dataloader)
        # would be a GPU fault or a silently wrong loss
        for nm, tns in (('target', target), ('valid', valid)):
            if tns is None:
                continue
            if not torch.is_tensor(tns) or not tns.is_cuda or tns.device != p.device:
                raise RuntimeError('gator_amd.train.losses: %s must be a tensor on %s (got %s)' % (nm, p.device, getattr(tns, 'device', type(tns))))
            if tns.dtype != torch.float32:
                raise TypeError('gator_amd.train.losses: %s must be float32 (got %s); convert it once when the batch is staged' % (nm, tns.dtype))
        if tuple(target.shape) != tuple(p.shape):
            raise ValueError('gator_amd.train.losses: target %s does not match the prediction %s' % (tuple(target.shape), tuple(p.shape)))
        t = target.contiguous()
        out = torch.empty(1, device=p.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = ops.zeros(p.shape, p.device) if need else None
        st = ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
        lib = _lib.load()
        if kind == 'coord':
            n4 = [1] * (4 - p.dim()) + list(p.shape)
            sv = None
            if valid is not None:
                ve = valid.expand(p.shape)
                sv = _I64x4(*([0] * (4 - p.dim()) + list(ve.stride())))
            ws = owner.workspace(max(1, p.numel() // 9 + 1), 1)
            _lib.check(lib.gator_t_coord_loss(p.data_ptr(), t.data_ptr(), valid.data_ptr() if valid is not None else None, sv, _I64x4(*n4), float(weight),
                                              out.data_ptr(), grad.data_ptr() if need else None, ws.data_ptr(), st), 'gator_t_coord_loss')
        else:
            B, V, _ = p.shape
            F = owner.faces.shape[0]
            ws = owner.workspace(B, F)
            fn = lib.gator_t_normal_loss if kind == 'normal' else lib.gator_t_edge_loss
            _lib.check(fn(p.data_ptr(), t.data_ptr(), owner.faces.data_ptr(), owner.inc_ptr.data_ptr(), owner.inc_idx.data_ptr(), B, V, F, float(weight),
                          out.data_ptr(), grad.data_ptr() if need else None, ws.data_ptr(), st), 'gator_t_%s_loss' % kind)
        ctx.save_for_backward(grad)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return ops.raw_binary(ops.MUL, grad, g.reshape([1] * grad.dim())), None, None, None, None, None


class MeshLosses:
    """get_loss(faces) of lib/core/loss.py:115-118 plus the weighting of Trainer.train (lib/core/base.py:137-148)."""

    def __init__(self, faces, j_regressor, device, num_verts=6890, normal_weight=1e-1, edge_weight=20.0, joint_weight=1e-3):
        f = np.asarray(faces, np.int32).reshape(-1, 3)
        if f.shape[0] < 1 or f.min() < 0 or f.max() >= num_verts:
            raise ValueError('MeshLosses: faces must hold at least one triangle with vertex indices in [0, %d)' % num_verts)
        ptr, idx = vertex_incidence(f, num_verts)
        self.faces = torch.from_numpy(np.ascontiguousarray(f)).to(device)
        self.inc_ptr = torch.from_numpy(ptr).to(device)
        self.inc_idx = torch.from_numpy(idx).to(device)
        self.j_regressor = torch.as_tensor(np.asarray(j_regressor, np.float32)).to(device)       # target_joint_set regressor, base.py:104
        self.normal_weight, self.edge_weight, self.joint_weight = normal_weight, edge_weight, joint_weight
        self._ws = None

    def workspace(self, B, F):
        n = int(_lib.load().gator_t_loss_ws_bytes(B, F))
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, device=self.faces.device, dtype=torch.uint8)
        return self._ws

    def coord(self, pred, target, valid=None, weight=1.0):
        return _LossFn.apply(pred, 'coord', weight, self, target, valid)

    def normal(self, pred, target, weight=1.0):
        return _LossFn.apply(pred, 'normal', weight, self, target, None)

    def edge(self, pred, target, weight=1.0):
        return _LossFn.apply(pred, 'edge', weight, self, target, None)

    def total(self, pred_mesh, lift_pose, targets, with_edge=False):
        """lib/core/base.py:137-148.  targets: dict with mesh [B,6890,3] m, reg_pose3d [B,Jr,3] mm, lift_pose3d [B,J,3] mm and their
        *_valid masks.  Returns (loss, parts)."""
        m1, m2, m3, m4 = ops.fork(pred_mesh, 4)
        pred_pose = ops.matmul(self.j_regressor, ops.affine(m1, 1000.0))
        parts = {
            'vertice': self.coord(m2, targets['mesh'], targets.get('mesh_valid')),
            'normal': self.normal(m3, targets['mesh'], self.normal_weight),
            'mesh2joint3d': self.coord(pred_pose, targets['reg_pose3d'], targets.get('reg_pose3d_valid'), self.joint_weight),
            'liftedjoint3d': self.coord(lift_pose, targets['lift_pose3d'], targets.get('lift_pose3d_valid'), self.joint_weight),
        }
        loss = ops.add(ops.add(parts['vertice'], parts['normal']), ops.add(parts['mesh2joint3d'], parts['liftedjoint3d']))
        if with_edge:
            parts['edge'] = self.edge(m4, targets['mesh'], self.edge_weight)
            loss = ops.add(loss, parts['edge'])
        return loss, parts
