"""Drop-in for the reference's lib/models/GATOR.py: GAT pose lifter -> concat -> MDR mesh regressor, one HIP forward."""
import torch

from .. import _lib
from ..config import cfg
from . import GAT, MDR
from ._base import HipModule


class GATOR(HipModule):
    PARTS = _lib.PART_GAT | _lib.PART_MDR

    def __init__(self, num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor, base_data=None, alpha=None):
        super().__init__()
        self.num_joint = num_joint
        self.precision = 'f32'        # 'bf16': vertex regressor on bf16 MFMA (BASELINE config 3), the rest stays fp32
        self.pose_lifter = GAT.get_model(num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor,
                                         pretrained=cfg.MODEL.posenet_pretrained, base_data=base_data)   # GATOR.py:13
        self.pose2mesh = MDR.get_model(num_joint, embed_dim, base_data=base_data, alpha=alpha)           # GATOR.py:14

    def _const_tensors(self):
        d = dict(self.pose_lifter._const_tensors())
        d.update(self.pose2mesh._const_tensors())
        return d

    def _config(self):
        return {'num_joint': self.num_joint, 'alpha': self.pose2mesh.alpha}

    supports_out = True        # forward(x, out=(verts, pose3d)) writes into the caller's buffers (ShardedForward: its rank's slice of the gather buffer)

    def forward(self, pose2d, out=None):
        """pose2d [B,J,2] -> (cam_mesh [B,6890,3] metres, pose3d [B,J,3] mm);  lib/models/GATOR.py:16-22.
        `out` = (verts, pose3d): contiguous fp32 tensors of those shapes on the input's device that the kernels write instead of
        fresh allocations."""
        x = self._prep(pose2d, 'GATOR.forward')
        B = x.shape[0]
        if out is not None:
            verts, pose3d = out
            for t, shp in ((verts, (B, 6890, 3)), (pose3d, (B, self.num_joint, 3))):
                if tuple(t.shape) != shp or t.dtype != torch.float32 or t.device != x.device or not t.is_contiguous():
                    raise RuntimeError('GATOR.forward: out tensors must be contiguous fp32 %s on %s' % (shp, x.device))
        else:
            verts = torch.empty((B, 6890, 3), device=x.device, dtype=torch.float32)
            pose3d = torch.empty((B, self.num_joint, 3), device=x.device, dtype=torch.float32)
        def call(ctx):       # the 16-bit mode exists on the default operand forms only: a module healed to 'exact' (HipModule._run) runs fp32
            lib = _lib.load()
            fn = lib.gator_forward_bf16 if (self.precision == 'bf16' and self.arithmetic == 'default') else lib.gator_forward_f32
            return fn(ctx, x.data_ptr(), B, verts.data_ptr(), pose3d.data_ptr(), self._stream(x.device))
        self._run(x.device, call, 'gator_forward_' + self.precision, outputs=(verts, pose3d))
        return verts, pose3d


    def set_joint_regressor(self, dense):
        """Register a [n_joint, 6890] joint regressor (lib/core/base.py:221, demo/run.py:142) for forward_joints()."""
        import numpy as np
        d = np.asarray(dense, np.float32)
        r, c = np.nonzero(d)
        self._jreg = (np.ascontiguousarray(r.astype(np.int32)), np.ascontiguousarray(c.astype(np.int32)),
                      np.ascontiguousarray(d[r, c].astype(np.float32)), int(d.shape[0]))
        self._jreg_ctx = None

    def forward_joints(self, pose2d, with_verts=False):
        """GATOR.forward + J_regressor @ mesh fused (gator_forward_joints_f32): -> (joints [B,n_joint,3] metres, pose3d [B,J,3] mm
        [, verts]).  Without `with_verts` no vertex is written at all."""
        if getattr(self, '_jreg', None) is None:
            raise RuntimeError('call set_joint_regressor() first')
        x = self._prep(pose2d, 'GATOR.forward_joints')
        B = x.shape[0]
        lib = _lib.load()
        r, c, v, nj = self._jreg
        joints = torch.empty((B, nj, 3), device=x.device, dtype=torch.float32)
        pose3d = torch.empty((B, self.num_joint, 3), device=x.device, dtype=torch.float32)
        verts = torch.empty((B, 6890, 3), device=x.device, dtype=torch.float32) if with_verts else None

        def call(ctx):       # the regressor belongs to a context: (re-)registered when the context is new (first call, healed arithmetic)
            if getattr(self, '_jreg_ctx', None) != ctx.value:
                _lib.check(lib.gator_set_joint_regressor(ctx, r.ctypes.data, c.ctypes.data, v.ctypes.data, int(r.size), nj), 'gator_set_joint_regressor')
                self._jreg_ctx = ctx.value
            return lib.gator_forward_joints_f32(ctx, x.data_ptr(), B, joints.data_ptr(), pose3d.data_ptr(),
                                                verts.data_ptr() if with_verts else None, self._stream(x.device))
        res = (joints, pose3d, verts) if with_verts else (joints, pose3d)
        self._run(x.device, call, 'gator_forward_joints_f32', outputs=res)
        return res


def get_model(num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor, **kw):
    return GATOR(num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor, **kw)
