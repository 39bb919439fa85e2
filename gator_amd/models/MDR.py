"""Drop-in for the reference's lib/models/MDR.py (motion-disentangled regression head): same factory and state_dict
layout; forward executed by the HIP library.  Constructor constants follow lib/models/MDR.py:72-90."""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..config import cfg
from ._base import HipModule, load_base_data


class MDR(HipModule):
    PARTS = _lib.PART_MDR

    def __init__(self, num_joint, embed_dim, base_data=None, alpha=None):
        super().__init__()
        if embed_dim != 128:
            raise ValueError('libgator_hip is built for embed_dim=128 (lib/core/base.py:57)')
        self.embed_dim, self.num_joint = 64, num_joint                      # MDR.py:74-75 (width hard-coded to 64)
        self.alpha = bool(cfg.MODEL.alpha if alpha is None else alpha)      # MDR.py:115
        base = load_base_data(base_data)
        mean_v = np.ascontiguousarray(base['smpl_mean_vertices'], np.float32)
        v = mean_v
        for d in base['D']:                                                 # Mesh.downsample 6890 -> 1723 -> 431, MDR.py:80-81
            v = np.asarray(d.astype(np.float32) @ v, np.float32)
        v431 = np.ascontiguousarray(v)
        self.num_verts = v431.shape[0]
        if self.num_verts != 431:
            raise ValueError('mesh_downsampling must coarsen 6890 -> 431 vertices, got %d' % self.num_verts)
        jr = torch.from_numpy(np.asarray(base['J_regressor_h36m'], np.float32))      # ALWAYS the h36m regressor, MDR.py:85
        self.joints_template = torch.matmul(jr, torch.from_numpy(mean_v))
        jt = np.ascontiguousarray(self.joints_template.numpy(), np.float32)
        rel = np.zeros(self.num_verts, np.int32)
        _lib.check(_lib.load().gator_verts_joints_relation(jt.ctypes.data, jt.shape[0], v431.ctypes.data, self.num_verts,
                                                          rel.ctypes.data), 'gator_verts_joints_relation')
        self.vj_relation = rel
        J, E, V = num_joint, 64, self.num_verts
        D = self._declare
        D('init_vertices', (V, 3), None, buffer=True, value=torch.from_numpy(v431))
        D('init_vertices_6890', (6890, 3), None, buffer=True, value=torch.from_numpy(mean_v))
        D('pos_j_id_embed.weight', (J + 1, E), 'embed'); D('pos_v_id_embed.weight', (V + 1, E), 'embed')
        for sfx in ('', '_1', '_2'):
            e = 'encoder%s.' % sfx
            D(e + 'norm1.weight', (E,), 'ones'); D(e + 'norm1.bias', (E,), 'zeros')
            for n in ('wq', 'wk', 'wv'):
                D(e + 'attn.%s.weight' % n, (E, E), 'linear_w')
            D(e + 'attn.proj.weight', (E, E), 'linear_w'); D(e + 'attn.proj.bias', (E,), 'bias:%d' % E)
            D(e + 'norm2.weight', (E,), 'ones'); D(e + 'norm2.bias', (E,), 'zeros')
            D(e + 'mlp.fc1.weight', (4 * E, E), 'linear_w'); D(e + 'mlp.fc1.bias', (4 * E,), 'bias:%d' % E)
            D(e + 'mlp.fc2.weight', (E, 4 * E), 'linear_w'); D(e + 'mlp.fc2.bias', (E,), 'bias:%d' % (4 * E))
            for n in range(4):
                D('selfatt%s.linears.%d.weight' % (sfx, n), (E, E), 'linear_w')
                D('selfatt%s.linears.%d.bias' % (sfx, n), (E,), 'bias:%d' % E)
            D('norm%s.a_2' % sfx, (E,), 'ones'); D('norm%s.b_2' % sfx, (E,), 'zeros')
        D('get_joint_feature.weight', (E, 2 + 3 + embed_dim), 'linear_w'); D('get_joint_feature.bias', (E,), 'bias:133')
        D('get_verts_feature.weight', (E, 6), 'linear_w'); D('get_verts_feature.bias', (E,), 'bias:6')
        D('motion_linear.weight', (23, E), 'linear_w'); D('motion_linear.bias', (23,), 'bias:%d' % E)
        D('bias_linear.weight', (3, E), 'linear_w'); D('bias_linear.bias', (3,), 'bias:%d' % E)
        if self.alpha:                                                      # MDR.py:115-117
            D('bias_norm.weight', (3,), 'ones'); D('bias_norm.bias', (3,), 'zeros')
            D('scale_linear.weight', (1, E), 'linear_w'); D('scale_linear.bias', (1,), 'bias:%d' % E)
        else:                                                               # MDR.py:119 BatchNorm1d(431)
            D('bias_norm.weight', (V,), 'ones'); D('bias_norm.bias', (V,), 'zeros')
            D('bias_norm.running_mean', (V,), 'zeros', buffer=True); D('bias_norm.running_var', (V,), 'ones', buffer=True)
            D('bias_norm.num_batches_tracked', (), None, buffer=True, value=torch.tensor(0, dtype=torch.long))
        D('bias_conv1d.weight', (20, V, 3), 'linear_w'); D('bias_conv1d.bias', (20,), 'bias:%d' % (3 * V))
        D('upsample_conv.weight', (6890, V, 3), 'linear_w'); D('upsample_conv.bias', (6890,), 'bias:%d' % (3 * V))

    def _const_tensors(self):
        return {'const.vj_relation': self.vj_relation}

    def _config(self):
        return {'num_joint': self.num_joint, 'alpha': self.alpha}

    def _context(self, device):
        # num_batches_tracked (int64 scalar) is part of the checkpoint but not of the arithmetic
        return super()._context(device)

    def forward(self, x):
        """x = cat(pose2d, pose3d/1000, feat) [B,J,133] -> vertices [B,6890,3] (m);  lib/models/MDR.py:124-170."""
        x = self._prep(x, 'MDR.forward')
        B = x.shape[0]
        verts = torch.empty((B, 6890, 3), device=x.device, dtype=torch.float32)
        self._run(x.device, lambda ctx: _lib.load().gator_mdr_forward_f32(ctx, x.data_ptr(), B, verts.data_ptr(), self._stream(x.device)),
                  'gator_mdr_forward_f32', outputs=verts)
        return verts

    def upsample(self, vert431, precision='f32'):
        """upsample_conv + template add alone (MDR.py:167-168): [B,431,3] -> [B,6890,3]."""
        x = self._prep(vert431, 'MDR.upsample')
        B = x.shape[0]
        verts = torch.empty((B, 6890, 3), device=x.device, dtype=torch.float32)
        fn = _lib.load().gator_upsample_bf16 if precision == 'bf16' else _lib.load().gator_upsample_f32
        self._run(x.device, lambda ctx: fn(ctx, x.data_ptr(), B, verts.data_ptr(), self._stream(x.device)), 'gator_upsample_' + precision, outputs=verts)
        return verts


def get_model(num_joint, embed_dim, **kw):
    return MDR(num_joint, embed_dim, **kw)
