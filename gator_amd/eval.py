"""Caller-side steps that FOLLOW the hot path in the reference's eval loop (SURVEY 8f "next" rows), kept on the GPU:

  joints  = J_regressor @ (verts * 1000)          lib/core/base.py:219-221, demo/run.py:142   -> gator_regress_joints_f32
  MPJPE   = mean || root-aligned pred - target ||  data/PW3D/dataset.py:273-286, data/Human36M/dataset.py:466-478
  PA-MPJPE via the rigid (Procrustes) alignment    lib/coord_utils.py:127-149

so that evaluation needs [B,17,3] + a few scalars on the host instead of a D2H copy of every 82 kB mesh
(lib/core/base.py:223,232-237 copy each mesh to the host twice)."""
import ctypes

import numpy as np
import torch

from . import _lib

H36M_EVAL_JOINTS = (1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15, 16)     # data/PW3D/dataset.py:46


class JointRegressor:
    """Sparse [n_joint, 6890] regressor held on the device as COO (107 / 105 non-zeros for the shipped regressors)."""

    def __init__(self, dense, device):
        d = np.asarray(dense, np.float32)
        r, c = np.nonzero(d)
        self.n_joint = int(d.shape[0])
        self.nnz = int(r.size)
        self.row = torch.from_numpy(r.astype(np.int32)).to(device)
        self.col = torch.from_numpy(c.astype(np.int32)).to(device)
        self.val = torch.from_numpy(d[r, c].astype(np.float32)).to(device)

    def __call__(self, verts):
        """verts [B,6890,3] f32 on the device -> joints [B,n_joint,3] (same unit as verts)."""
        if not verts.is_cuda:
            raise RuntimeError('JointRegressor: verts must live on a HIP device')
        verts = verts.contiguous().float()
        B = verts.shape[0]
        out = torch.empty((B, self.n_joint, 3), device=verts.device, dtype=torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream(verts.device).cuda_stream)
        _lib.check(_lib.load().gator_regress_joints_f32(verts.data_ptr(), B, self.row.data_ptr(), self.col.data_ptr(),
                                                       self.val.data_ptr(), self.nnz, self.n_joint, out.data_ptr(), st),
                   'gator_regress_joints_f32')
        return out


def mpjpe(pred_joint, target_joint, eval_joints=H36M_EVAL_JOINTS, root=0):
    """Root-align, select the evaluation joints, mean Euclidean distance (data/PW3D/dataset.py:273-286).  Device tensors."""
    p = pred_joint - pred_joint[:, root:root + 1]
    t = target_joint - target_joint[:, root:root + 1]
    if eval_joints is not None:
        idx = torch.as_tensor(eval_joints, device=p.device)
        p, t = p[:, idx], t[:, idx]
    return torch.sqrt(((p - t) ** 2).sum(2)).mean()


def mpvpe(pred_mesh, target_mesh, pred_joint, target_joint, root=0):
    """Mesh error after aligning both meshes to their own root joint (data/PW3D/dataset.py:275,283)."""
    p = pred_mesh - pred_joint[:, root:root + 1]
    t = target_mesh - target_joint[:, root:root + 1]
    return torch.sqrt(((p - t) ** 2).sum(2)).mean()


def rigid_align(a, b):
    """Batched similarity (Procrustes) alignment of a onto b, [B,N,3] device tensors (lib/coord_utils.py:127-149:
    rotation by SVD of the covariance with the reflection fix, scale = trace(S)/var(a), translation of the centroids)."""
    ca, cb = a.mean(1, keepdim=True), b.mean(1, keepdim=True)
    a0, b0 = a - ca, b - cb
    H = a0.transpose(1, 2) @ b0
    U, S, Vt = torch.linalg.svd(H.double())
    R = Vt.transpose(1, 2) @ U.transpose(1, 2)
    neg = torch.linalg.det(R) < 0
    Vt = torch.where(neg[:, None, None], torch.cat([Vt[:, :2], -Vt[:, 2:]], 1), Vt)
    S = torch.where(neg[:, None], torch.cat([S[:, :2], -S[:, 2:]], 1), S)
    R = Vt.transpose(1, 2) @ U.transpose(1, 2)
    var_a = (a0.double() ** 2).sum((1, 2)) / a.shape[1]
    c = (S.sum(1) / a.shape[1]) / var_a
    t = cb.double().transpose(1, 2) - c[:, None, None] * (R @ ca.double().transpose(1, 2))
    return (c[:, None, None] * (R @ a.double().transpose(1, 2)) + t).transpose(1, 2).to(a.dtype)


def pa_mpjpe(pred_joint, target_joint, eval_joints=H36M_EVAL_JOINTS):
    idx = torch.as_tensor(eval_joints, device=pred_joint.device)
    p, t = pred_joint[:, idx], target_joint[:, idx]
    return torch.sqrt(((rigid_align(p, t) - t) ** 2).sum(2)).mean()
