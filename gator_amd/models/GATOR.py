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

    def forward(self, pose2d):
        """pose2d [B,J,2] -> (cam_mesh [B,6890,3] metres, pose3d [B,J,3] mm);  lib/models/GATOR.py:16-22."""
        x = self._prep(pose2d, 'GATOR.forward')
        B = x.shape[0]
        ctx = self._context(x.device)
        verts = torch.empty((B, 6890, 3), device=x.device, dtype=torch.float32)
        pose3d = torch.empty((B, self.num_joint, 3), device=x.device, dtype=torch.float32)
        fn = _lib.load().gator_forward_bf16 if self.precision == 'bf16' else _lib.load().gator_forward_f32
        _lib.check(fn(ctx, x.data_ptr(), B, verts.data_ptr(), pose3d.data_ptr(), self._stream(x.device)), 'gator_forward_' + self.precision)
        return verts, pose3d


def get_model(num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor, **kw):
    return GATOR(num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor, **kw)
