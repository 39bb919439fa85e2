"""Drop-in for the reference's lib/models/GAT.py: same factory, same state_dict keys/shapes, forward
executed by the HIP library.  Constructor constants follow lib/models/GAT.py:56-112 (adjacency surgery, template-joint edge
lengths, hop/path tensors); the Floyd-Warshall / path expansion that the reference takes from absent Cython output is
computed by libgator_hip's host helpers when the .npy files are not supplied."""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from ..config import cfg
from ._base import HipModule, load_base_data


def _dense_adj(graph_adj):
    a = graph_adj[-1]
    a = a.toarray() if hasattr(a, 'toarray') else np.asarray(a)
    return np.ascontiguousarray(a, dtype=np.float32)


class GAT(HipModule):
    PARTS = _lib.PART_GAT

    def __init__(self, num_joint=17, embed_dim=256, depth=4, graph_adj=None, GCN_depth=1, J_regressor=None, num_heads=8,
                 mlp_ratio=4., qkv_bias=True, pretrained=False, base_data=None, **_ignored):
        super().__init__()
        if embed_dim != 128 or depth != 6 or num_heads != 8:
            raise ValueError('libgator_hip is built for the configuration every reference entry point uses: '
                             'embed_dim=128, depth=6, num_heads=8 (lib/core/base.py:57, demo/run.py:96)')
        if num_joint not in (17, 19):
            raise ValueError('num_joint must be 17 or 19 (lib/models/GAT.py:79-93)')
        if graph_adj is None or J_regressor is None:
            raise ValueError('graph_adj and J_regressor are required (lib/models/GAT.py:57-76)')
        J, C = num_joint, embed_dim
        self.num_joint, self.embed_dim, self.num_heads, self.output_size = J, C, num_heads, 3 * J
        base = load_base_data(base_data, J)
        # --- topology: dense adjacency with the hard-coded h36m symmetric-edge deletions (GAT.py:58-65) ---
        adj = _dense_adj(graph_adj)
        for i, j in ((1, 4), (2, 5), (3, 6), (11, 14), (12, 15), (13, 16)):
            adj[i, j] = adj[j, i] = 0
        mean_v = torch.from_numpy(np.asarray(base['smpl_mean_vertices'])).unsqueeze(0)
        jreg = torch.as_tensor(J_regressor, dtype=torch.float32)
        tj = torch.matmul(jreg[None], mean_v.float()).squeeze(0)                       # GAT.py:76
        if J == 19:                                                                    # GAT.py:79-88
            tj = torch.cat((tj, ((tj[11] + tj[12]) * 0.5)[None], ((tj[5] + tj[6]) * 0.5)[None]), 0)
        lib = _lib.load()
        sp, path = base.get('shortest_path'), base.get('path')
        if sp is None or path is None:
            sp, path = np.zeros((J, J), np.int64), np.zeros((J, J), np.int64)
            _lib.check(lib.gator_floyd_warshall(adj.ctypes.data, J, sp.ctypes.data, path.ctypes.data), 'gator_floyd_warshall')
        sp, path = np.ascontiguousarray(sp, np.int64), np.ascontiguousarray(path, np.int64)
        ed = np.zeros((J, J), np.float32)                                              # upper-triangular, GAT.py:95-108
        for i in range(J):
            for j in range(i + 1, J):
                if adj[i, j] == 1:
                    ed[i, j] = math.sqrt(((tj[i] - tj[j]) ** 2).sum(0))
        max_dist = int(sp.max())
        edge_input = np.zeros((J, J, max_dist), np.float32)
        _lib.check(lib.gator_gen_edge_input(path.ctypes.data, ed.ctypes.data, J, max_dist, edge_input.ctypes.data),
                   'gator_gen_edge_input')
        self.spatial_pos, self.path, self.edge_input = sp, path, edge_input           # plain attributes, as in the reference
        # --- state_dict layout (Appendix B) ---
        D = self._declare
        D('graph_adj', (J, J), None, buffer=True, value=torch.from_numpy(adj))
        D('pos_id_embed.weight', (J + 1, C), 'embed')
        D('GLinear.0.W', (64, 2), 'uniform:%g' % (1 / 128)); D('GLinear.0.b', (64,), 'uniform:%g' % (1 / 128))
        D('GLinear.1.weight', (64,), 'ones'); D('GLinear.1.bias', (64,), 'zeros')
        D('GLinear.3.W', (C, 64), 'uniform:%g' % (1 / (64 * C))); D('GLinear.3.b', (C,), 'uniform:%g' % (1 / (64 * C)))
        D('pos_num_embed.weight', (J, C), 'embed')
        D('init_vertices', (1, 6890, 3), None, buffer=True, value=mean_v)
        D('get_hop_path_encoding.W', (8, J, J, max_dist), 'ones')
        D('get_hop_path_encoding.spatial_pos_encoder.weight', (10, 8), 'embed')
        D('get_hop_path_encoding.edge_encoder.weight', (8 * J * J, J * J), 'linear_w')
        D('get_hop_path_encoding.edge_encoder.bias', (8 * J * J,), 'bias:%d' % (J * J))
        for i in range(depth):
            b = 'blocks.%d.' % i
            D(b + 'norm1.weight', (C,), 'ones'); D(b + 'norm1.bias', (C,), 'zeros')
            D(b + 'attn.qkv.weight', (3 * C, C), 'linear_w'); D(b + 'attn.qkv.bias', (3 * C,), 'bias:%d' % C)
            D(b + 'attn.proj.weight', (C, C), 'linear_w'); D(b + 'attn.proj.bias', (C,), 'bias:%d' % C)
            D(b + 'norm2.weight', (C,), 'ones'); D(b + 'norm2.bias', (C,), 'zeros')
            D(b + 'mlp.fc1.weight', (4 * C, C), 'linear_w'); D(b + 'mlp.fc1.bias', (4 * C,), 'bias:%d' % C)
            D(b + 'mlp.fc2.weight', (C, 4 * C), 'linear_w'); D(b + 'mlp.fc2.bias', (C,), 'bias:%d' % (4 * C))
            D(b + 'gcn.W', (2, C, C), 'uniform:%g' % (1.414 * math.sqrt(6.0 / (C * C + 2 * C))))
            D(b + 'gcn.M', (J, C), 'uniform:%g' % (1.414 * math.sqrt(6.0 / (J + C))))
            D(b + 'gcn.adj2', (J, J), 'const:1e-6'); D(b + 'gcn.bias', (C,), 'bias:%d' % C)
            D(b + 'x_feat.linears.0.weight', (C, C), 'linear_w'); D(b + 'x_feat.linears.0.bias', (C,), 'bias:%d' % C)
            D(b + 'x_feat.linears.1.weight', (C // 8, C), 'linear_w'); D(b + 'x_feat.linears.1.bias', (C // 8,), 'bias:%d' % C)
            D(b + 'x_feat.linearback.weight', (C, C + C // 8), 'linear_w')
            D(b + 'x_feat.linearback.bias', (C,), 'bias:%d' % (C + C // 8))
        D('norm.weight', (C,), 'ones'); D('norm.bias', (C,), 'zeros')
        D('lifter.weight', (3 * J, C * J), 'linear_w'); D('lifter.bias', (3 * J,), 'bias:%d' % (C * J))
        if pretrained:
            self._load_pretrained_model()

    def _load_pretrained_model(self):   # lib/models/GAT.py:128-131
        import glob
        import os
        files = sorted(glob.glob(os.path.join(cfg.MODEL.posenet_path, '*.pth.tar')))
        if not files:
            raise ValueError('No checkpoint exists!\n', cfg.MODEL.posenet_path)   # lib/funcs_utils.py:121-127
        best = [f for f in files if f.endswith('best.pth.tar')]
        ck = torch.load(best[0] if best else files[-1], map_location='cpu')
        self.load_state_dict(ck['model_state_dict'])

    def _const_tensors(self):
        return {'const.shortest_path': self.spatial_pos, 'const.edge_input': self.edge_input}

    def _config(self):
        return {'num_joint': self.num_joint, 'alpha': False}

    def forward(self, pose2d):
        """pose2d [B, 2J] (or [B,J,2]) -> (x_out [B,3J] mm, feat [B,J,C]);  lib/models/GAT.py:133-152."""
        x = self._prep(pose2d, 'GAT.forward')
        B = x.shape[0]
        x_out = torch.empty((B, 3 * self.num_joint), device=x.device, dtype=torch.float32)
        feat = torch.empty((B, self.num_joint, self.embed_dim), device=x.device, dtype=torch.float32)
        self._run(x.device, lambda ctx: _lib.load().gator_gat_forward_f32(ctx, x.data_ptr(), B, x_out.data_ptr(), feat.data_ptr(),
                                                                         self._stream(x.device)), 'gator_gat_forward_f32', outputs=(x_out, feat))
        return x_out, feat


def get_model(num_joint=17, embed_dim=256, depth=4, graph_adj=None, GCN_depth=1, J_regressor=None, pretrained=False, **kw):
    return GAT(num_joint, embed_dim, depth, graph_adj, GCN_depth, J_regressor, pretrained=pretrained, **kw)
