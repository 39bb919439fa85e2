"""Seeded synthetic stand-ins for the licence-gated base data and for trained weights.

The reference needs files that are not redistributable / not shipped (README.md:47-56,97-107 of the
reference): ``smpl_mean_vertices.npy``, ``mesh_downsampling.npz``, ``J_regressor_h36m.npy``, the
``*.pth.tar`` checkpoints.  For parity tests and benchmarking the *arithmetic* is what matters, so
both sides (reference import in tools/gen_golden.py, this package, the oracle) regenerate identical
stand-ins from ``np.random.RandomState(seed)`` (legacy stream, frozen across NumPy versions) instead
of committing 46 MB of weights.  Recipe: SURVEY.md Appendix F.
"""
import os
import re

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_JREG_FIXTURE = os.path.join(os.path.dirname(_HERE), 'tests', 'golden', 'j_regressors.npz')


def load_j_regressors(path=_JREG_FIXTURE):
    """The two joint regressors the reference ships (data/Human36M/J_regressor_h36m_correct.npy,
    data/COCO/J_regressor_coco.npy), stored as COO.  Falls back to a seeded sparse stand-in."""
    out = {}
    if os.path.exists(path):
        z = np.load(path)
        for name in ('h36m', 'coco'):
            m = np.zeros((17, 6890), np.float64)
            m[z[name + '_row'], z[name + '_col']] = z[name + '_val']
            out[name] = m
        return out
    rs = np.random.RandomState(12345)
    for name in ('h36m', 'coco'):
        m = np.zeros((17, 6890), np.float64)
        for j in range(17):
            cols = rs.choice(6890, 6, replace=False)
            w = rs.rand(6) + 0.1
            m[j, cols] = w / w.sum()
        out[name] = m
    return out


def make_base_data(seed=0):
    """smpl_mean_vertices, D[0] (1723x6890), D[1] (431x1723), J_regressor_h36m -- drawn first from the stream."""
    rs = np.random.RandomState(seed)
    mean_v = (0.3 * rs.randn(6890, 3)).astype(np.float32)

    def row_select(n_out, n_in):
        cols = rs.permutation(n_in)[:n_out]
        return sp.csr_matrix((np.ones(n_out, np.float32), (np.arange(n_out), cols)), shape=(n_out, n_in))

    d0 = row_select(1723, 6890)
    d1 = row_select(431, 1723)
    jr = load_j_regressors()
    return {'smpl_mean_vertices': mean_v, 'D': [d0, d1],
            'J_regressor_h36m': jr['h36m'].astype(np.float32), 'rs': rs}


def model_j_regressor(num_joint):
    """The regressor a reference caller passes to get_model (lib/core/base.py:53, demo/run.py:69,80)."""
    jr = load_j_regressors()
    return (jr['h36m'] if num_joint == 17 else jr['coco']).astype(np.float32)


def _fan_in(key, shape):
    if key.endswith('gcn.W'):
        return shape[-2]
    return int(np.prod(shape[1:]))


def seeded_state_dict(shapes, rs, keep=(), upsample_gain=0.2):
    """shapes: {key: (shape tuple, is_int)} of a reference-layout state_dict.  Returns {key: ndarray}
    for every key not in ``keep`` (buffers derived from base data keep their constructed values)."""
    out = {}
    for key in sorted(shapes):
        shape, is_int = shapes[key]
        leaf = key.rsplit('.', 1)[-1]
        if key in keep or leaf in ('graph_adj', 'init_vertices', 'init_vertices_6890'):
            continue
        if is_int:                                                   # num_batches_tracked
            out[key] = np.zeros(shape, np.int64)
        elif leaf == 'running_mean':
            out[key] = rs.randn(*shape).astype(np.float32)
        elif leaf == 'running_var':
            out[key] = rs.uniform(0.5, 2.0, size=shape).astype(np.float32)
        elif re.search(r'(pos_\w+_embed|spatial_pos_encoder)\.weight$', key):
            w = (0.02 * rs.randn(*shape)).astype(np.float32)
            w[0] = 0                                                 # padding_idx row
            out[key] = w
        elif leaf == 'M':
            out[key] = (1 + 0.3 * rs.randn(*shape)).astype(np.float32)
        elif leaf == 'adj2':
            out[key] = (0.05 * rs.randn(*shape)).astype(np.float32)
        elif key.endswith('get_hop_path_encoding.W'):
            out[key] = (1 + 0.3 * rs.randn(*shape)).astype(np.float32)
        elif leaf in ('weight', 'W') and len(shape) >= 2:
            # The output head gets gain 0.2 so that synthetic meshes have human-scale extent (rms 0.33 m,
            # max 1.6 m).  At gain 1 the vertices reach 5 m and the reference's OWN fp32-vs-fp64 noise is
            # 3.3e-3 mm, which would make the 1e-3 mm parity criterion unmeetable by the reference itself.
            # `upsample_gain` = 1.0 is kept as a second, scale-free parity point (tests/golden/scale_gain*.npz).
            gain = upsample_gain if key.endswith('upsample_conv.weight') else 1.0
            out[key] = (gain * rs.randn(*shape) / np.sqrt(_fan_in(key, shape))).astype(np.float32)
        elif leaf in ('weight', 'a_2'):                              # norm scales (1-D)
            out[key] = (1 + 0.1 * rs.randn(*shape)).astype(np.float32)
        elif leaf in ('bias', 'b', 'b_2'):
            out[key] = (0.02 * rs.randn(*shape)).astype(np.float32)
        else:
            raise KeyError('no seeded recipe for %s %s' % (key, shape))
    return out


def shapes_of(state_dict):
    return {k: (tuple(v.shape), not v.dtype.is_floating_point) for k, v in state_dict.items()}


def synthetic_pose2d(batch, num_joint, seed=0, jitter=None):
    """Input contract a0 (data/PW3D/dataset.py:244-250): per-sample, per-axis zero-mean / unit population std."""
    rs = np.random.RandomState(seed)
    x = rs.randn(batch, num_joint, 2)
    if jitter is not None:
        x = x + jitter * rs.randn(batch, num_joint, 2)
    x = (x - x.mean(1, keepdims=True)) / x.std(1, keepdims=True)
    return x.astype(np.float32)


def synthetic_faces(seed=0, num_faces=13776, num_verts=6890):
    """Stand-in for the licence-gated SMPL face list (lib/core/base.py:63 reads main_dataset.mesh_model.face): seeded random
    triangles with three distinct vertices each, every vertex used.  Same count as SMPL (13 776)."""
    rs = np.random.RandomState(seed + 5)
    a = rs.permutation(num_faces) % num_verts                     # every vertex appears at least once
    b = (a + 1 + rs.randint(0, num_verts - 1, num_faces)) % num_verts
    c = rs.randint(0, num_verts, num_faces)
    bad = (c == a) | (c == b)
    while bad.any():
        c[bad] = rs.randint(0, num_verts, int(bad.sum()))
        bad = (c == a) | (c == b)
    return np.stack([a, b, c], 1).astype(np.int32)


def training_targets(batch, num_joint, base, j_regressor_target, seed=0):
    """Synthetic stand-ins for one training batch's targets and masks (data/Human36M/dataset.py:393-405 layout): gt mesh =
    template + smooth noise (metres), regressed joints (mm), lifted pose (mm); one sample's mesh and one joint masked out."""
    rs = np.random.RandomState(seed + 11)
    mv = np.asarray(base['smpl_mean_vertices'], np.float32)
    mesh = (mv[None] * (1.0 + 0.05 * rs.randn(batch, 1, 3)) + 0.02 * rs.randn(batch, mv.shape[0], 3)).astype(np.float32)
    reg = (np.asarray(j_regressor_target, np.float32)[None] @ (mesh * 1000.0) + 5.0 * rs.randn(batch, j_regressor_target.shape[0], 3)).astype(np.float32)
    lift = (300.0 * rs.randn(batch, num_joint, 3)).astype(np.float32)
    mesh_valid = np.ones((batch, mv.shape[0], 1), np.float32)
    reg_valid = np.ones((batch, j_regressor_target.shape[0], 1), np.float32)
    lift_valid = np.ones((batch, num_joint, 1), np.float32)
    if batch > 1:
        mesh_valid[1] = 0                                          # a sample without a fitted mesh (dataset.py:399)
        lift_valid[batch - 1, 3] = 0
    return {'mesh': mesh, 'reg_pose3d': reg, 'lift_pose3d': lift, 'mesh_valid': mesh_valid, 'reg_pose3d_valid': reg_valid, 'lift_pose3d_valid': lift_valid}
