#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference (kasvii/GATOR,
mounted read-only at /root/reference) on CPU with import shims.  Dev-container only: the reference
never travels to the GPU box; only the small data fixtures written here do.

Usage:  python tools/gen_golden.py            (re-creates tests/golden/*.npz)

Shims: timm DropPath/Mlp stubs, easydict stub, a pre-seeded core.config.cfg (the real
one mkdirs under the read-only tree at import), funcs_utils stub (imports cv2/matplotlib), a 3-point
affine solve standing in for cv2.getAffineTransform, identity .cuda(), Mesh default device -> cpu,
chdir to a scratch dir holding seeded stand-ins for the licence-gated data/base_data files.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)

from gator_amd import synthetic  # noqa: E402
from oracle import graph_consts as gc  # noqa: E402  (Floyd-Warshall output = the path files fed to the reference)

OUT = os.path.join(REPO, 'tests', 'golden')


class _EasyDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def install_shims(scratch, alpha):
    # timm
    timm = types.ModuleType('timm')
    tm = types.ModuleType('timm.models')
    tl = types.ModuleType('timm.models.layers')
    tv = types.ModuleType('timm.models.vision_transformer')

    class DropPath(nn.Module):
        def __init__(self, p=0.):
            super().__init__()

        def forward(self, x):
            return x

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            super().__init__()
            out_features = out_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    tl.DropPath = DropPath
    tv.Mlp = Mlp
    sys.modules.update({'timm': timm, 'timm.models': tm, 'timm.models.layers': tl,
                        'timm.models.vision_transformer': tv})
    ed = types.ModuleType('easydict')
    ed.EasyDict = _EasyDict
    sys.modules['easydict'] = ed
    # core.config
    cfg = _EasyDict()
    cfg.DATASET = _EasyDict(BASE_DATA_DIR=os.path.join(scratch, 'data', 'base_data'))
    cfg.MODEL = _EasyDict(alpha=alpha, posenet_pretrained=False, posenet_path='', input_shape=(384, 288))
    core = types.ModuleType('core')
    cc = types.ModuleType('core.config')
    cc.cfg = cfg
    core.config = cc
    sys.modules.update({'core': core, 'core.config': cc})
    fu = types.ModuleType('funcs_utils')
    fu.load_checkpoint = lambda *a, **k: (_ for _ in ()).throw(ValueError('No checkpoint exists'))
    fu.stop = sys.exit
    sys.modules['funcs_utils'] = fu
    # cv2: only getAffineTransform is reached (lib/aug_utils.py:164-167)
    cv2 = types.ModuleType('cv2')

    def get_affine_transform(src, dst):
        a = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
        return np.linalg.solve(a, np.asarray(dst, np.float64)).T

    cv2.getAffineTransform = get_affine_transform
    sys.modules['cv2'] = cv2
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    for p in (os.path.join(REF, 'lib'),):
        if p not in sys.path:
            sys.path.insert(0, p)
    return cfg


def write_base_data(scratch, base, sp, path, tag):
    d = os.path.join(scratch, 'data', 'base_data')
    os.makedirs(d, exist_ok=True)
    np.save(os.path.join(d, 'smpl_mean_vertices.npy'), base['smpl_mean_vertices'])
    np.save(os.path.join(d, 'J_regressor_h36m.npy'), base['J_regressor_h36m'])
    import scipy.sparse as sps
    D = np.empty(2, dtype=object)
    D[0], D[1] = base['D'][0].tocoo(), base['D'][1].tocoo()
    U = np.empty(2, dtype=object)
    U[0], U[1] = base['D'][0].T.tocoo(), base['D'][1].T.tocoo()
    A = np.empty(3, dtype=object)
    A[0], A[1], A[2] = sps.identity(6890).tocoo(), sps.identity(1723).tocoo(), sps.identity(431).tocoo()
    np.savez(os.path.join(d, 'mesh_downsampling.npz'), A=A, U=U, D=D)
    np.save(os.path.join(d, 'shortest_path_%s.npy' % tag), sp)
    np.save(os.path.join(d, 'path_%s.npy' % tag), path)


def dump_jregressors():
    z = {}
    for name, f in (('h36m', 'data/Human36M/J_regressor_h36m_correct.npy'), ('coco', 'data/COCO/J_regressor_coco.npy')):
        m = np.load(os.path.join(REF, f))
        r, c = np.nonzero(m)
        z[name + '_row'], z[name + '_col'], z[name + '_val'] = r.astype(np.int32), c.astype(np.int32), m[r, c]
    np.savez(os.path.join(OUT, 'j_regressors.npz'), **z)


def run_variant(name, J, alpha, seed, B=4, upsample_gain=0.2, subset=None, out_name=None):
    import scipy.sparse as sps
    scratch = tempfile.mkdtemp(prefix='gator_golden_')
    for m in [k for k in sys.modules if k.split('.')[0] in ('models', 'graph_utils', 'coarsening', 'core', 'funcs_utils')]:
        del sys.modules[m]
    install_shims(scratch, alpha)
    base = synthetic.make_base_data(seed)
    rs = base['rs']
    skeleton, flips = gc.joint_setting(J)
    adj0 = gc.build_adj(J, skeleton, flips)
    sp, path = gc.floyd_warshall(gc.delete_symmetric_edges(adj0))
    tag = '3dpw' if J == 19 else 'h36m'
    write_base_data(scratch, base, sp, path, tag)
    os.chdir(scratch)
    import models  # noqa  (reference lib/models)
    from models.backbones import mesh as ref_mesh
    ref_mesh.Mesh.__init__.__defaults__ = ('data/base_data/mesh_downsampling.npz', 1, 1, torch.device('cpu'))
    import graph_utils
    ref_adj = graph_utils.build_adj(J, skeleton, flips)            # the reference's own adjacency builder
    assert np.array_equal(ref_adj, adj0)
    graph_adj = [None, sps.csr_matrix(ref_adj)]
    jreg = synthetic.model_j_regressor(J)
    model = models.GATOR.get_model(J, 128, 6, graph_adj, 1, torch.Tensor(jreg))
    model.eval()
    sd = model.state_dict()
    new = synthetic.seeded_state_dict(synthetic.shapes_of(sd), rs, upsample_gain=upsample_gain)
    sd.update({k: torch.from_numpy(v) for k, v in new.items()})
    model.load_state_dict(sd)
    pose2d = synthetic.synthetic_pose2d(B, J, seed + 1 if subset is None else seed + 17)
    x = torch.from_numpy(pose2d)

    taps = {}
    hooks = []

    def tap(mod, key, sel=lambda o: o):
        hooks.append(mod.register_forward_hook(lambda m, i, o: taps.__setitem__(key, sel(o).detach().clone())))

    tap(model.pose_lifter.get_hop_path_encoding, 'hop_path_bias')
    tap(model.pose_lifter, 'feat', lambda o: o[1])
    for i, blk in enumerate(model.pose_lifter.blocks):
        tap(blk, 'gat_block%d' % i, lambda o: o[0])
    tap(model.pose2mesh.selfatt, 'mdr_attn0')
    tap(model.pose2mesh.norm_2, 'mdr_norm2')
    tap(model.pose2mesh.upsample_conv, 'vert431_in')
    up_in = {}
    hooks.append(model.pose2mesh.upsample_conv.register_forward_hook(
        lambda m, i, o: up_in.__setitem__('vert431', i[0].detach().clone())))
    # verts tokens after the 3rd LBF layer = input of motion_linear
    hooks.append(model.pose2mesh.motion_linear.register_forward_hook(
        lambda m, i, o: taps.__setitem__('mdr_lbf2', i[0].detach().clone())))
    with torch.no_grad():
        verts, pose3d = model(x)
    for h in hooks:
        h.remove()
    # fp64 re-evaluation of the same weights (tolerance anchor): cast plain-attribute tensors by hand
    m64 = model.double()
    for blk in m64.pose_lifter.blocks:
        blk.adj = blk.adj.double()
        blk.gcn.adj = blk.gcn.adj.double()
    hp = m64.pose_lifter.get_hop_path_encoding
    hp.edg_adj = hp.edg_adj.double()
    hp.spatial = hp.spatial.double()
    with torch.no_grad():
        verts64, pose3d64 = m64(x.double())
    if subset is not None:
        # scale-free parity fixture: a fixed vertex subset of a larger batch, reference fp32 output and its fp64 anchor
        v32, v64 = verts.float().numpy()[:, subset], verts64.numpy()[:, subset]
        noise = np.abs(verts.double().numpy() - verts64.numpy()).max() * 1e3
        np.savez(os.path.join(OUT, out_name + '.npz'), variant=name, seed=np.int64(seed), upsample_gain=np.float64(upsample_gain),
                 pose2d=pose2d, vertex_subset=subset.astype(np.int32), verts_f64=v64,
                 ref32_minus_f64=(v32.astype(np.float64) - v64).astype(np.float32),
                 pose3d_f64=pose3d64.numpy(), ref32_noise_mm_all_vertices=np.float64(noise),
                 verts_absmax=np.float64(np.abs(verts64.numpy()).max()))
        print('%s: gain %.1f B=%d  |verts| max %.3f m  ref fp32-vs-fp64 max %.3e mm (all vertices), %.3e mm (subset)'
              % (out_name, upsample_gain, B, np.abs(verts64.numpy()).max(), noise, np.abs(v32 - v64).max() * 1e3))
        os.chdir(REPO)
        return
    gat = model.pose_lifter
    out = dict(
        seed=np.int64(seed), alpha=np.bool_(alpha), num_joint=np.int64(J),
        pose2d=pose2d, verts=verts.float().numpy(), pose3d=pose3d.float().numpy(),
        verts_f64=verts64.numpy(), pose3d_f64=pose3d64.numpy(),
        hop_path_bias=taps['hop_path_bias'].float().numpy(), feat=taps['feat'].float().numpy(),
        gat_block0=taps['gat_block0'].float().numpy(), gat_block5=taps['gat_block5'].float().numpy(),
        mdr_lbf2=taps['mdr_lbf2'].float().numpy(), vert431=up_in['vert431'].float().numpy(),
        graph_adj=gat.graph_adj.float().numpy(), shortest_path=sp, path=path,
        edge_input=hp.edg_adj.float().numpy(),
        vj_relation=np.asarray(model.pose2mesh.vj_relation, np.int64),
        init_vertices_431=sd['pose2mesh.init_vertices'].float().numpy(),
        state_dict_keys=np.array(sorted(sd.keys())),
        state_dict_shapes=np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())]),
        # spot checks of the seeded weights so a recipe drift is caught without storing 46 MB
        w_probe_lifter=sd['pose_lifter.lifter.weight'].float().numpy()[:2, :8],
        w_probe_upconv=sd['pose2mesh.upsample_conv.weight'].float().numpy()[:2, :3],
    )
    err = np.abs(out['verts'].astype(np.float64) - out['verts_f64']).max() * 1000
    print('%s: J=%d alpha=%s  |verts| max %.3f m  fp32-vs-fp64 max %.2e mm' % (name, J, alpha, np.abs(out['verts']).max(), err))
    np.savez(os.path.join(OUT, name + '.npz'), **out)
    os.chdir(REPO)


def train_golden(name, J, alpha, seed, B=4):
    """One training step's loss and gradients from the REAL reference: model.train() with every nn.Dropout at p = 0 (DropPath
    is the identity stub) - the reference's RNG stream cannot be reproduced elsewhere - so BatchNorm1d runs on batch statistics;
    loss = lib/core/base.py:137-148 incl. the edge term, criteria = lib/core/loss.py get_loss(faces) loaded from its file.
    Stores loss parts and, per parameter, the gradient's max / norm and 48 fixed entries, from the fp32 run and an fp64 run."""
    import importlib.util
    import scipy.sparse as sps
    scratch = tempfile.mkdtemp(prefix='gator_golden_')
    for m in [k for k in sys.modules if k.split('.')[0] in ('models', 'graph_utils', 'coarsening', 'core', 'funcs_utils')]:
        del sys.modules[m]
    install_shims(scratch, alpha)
    base = synthetic.make_base_data(seed)
    rs = base['rs']
    skeleton, flips = gc.joint_setting(J)
    adj0 = gc.build_adj(J, skeleton, flips)
    sp, path = gc.floyd_warshall(gc.delete_symmetric_edges(adj0))
    write_base_data(scratch, base, sp, path, '3dpw' if J == 19 else 'h36m')
    os.chdir(scratch)
    import models  # noqa
    from models.backbones import mesh as ref_mesh
    ref_mesh.Mesh.__init__.__defaults__ = ('data/base_data/mesh_downsampling.npz', 1, 1, torch.device('cpu'))
    spec = importlib.util.spec_from_file_location('ref_core_loss', os.path.join(REF, 'lib', 'core', 'loss.py'))
    ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_loss)
    graph_adj = [None, sps.csr_matrix(adj0)]
    model = models.GATOR.get_model(J, 128, 6, graph_adj, 1, torch.Tensor(synthetic.model_j_regressor(J)))
    sd = model.state_dict()
    new = synthetic.seeded_state_dict(synthetic.shapes_of(sd), rs)
    sd.update({k: torch.from_numpy(v) for k, v in new.items()})
    model.load_state_dict(sd)
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    faces = synthetic.synthetic_faces(seed)
    jreg_t = synthetic.load_j_regressors()['h36m'].astype(np.float32)          # target_joint_set regressor (base.py:104)
    pose2d = synthetic.synthetic_pose2d(B, J, seed + 3)
    tg = synthetic.training_targets(B, J, base, jreg_t, seed)
    crit = ref_loss.get_loss(faces=faces)

    def step(mdl, dt):
        t = {k: torch.from_numpy(v).to(dt) for k, v in tg.items()}
        mdl.zero_grad()
        pred_mesh, lift_pose = mdl(torch.from_numpy(pose2d).to(dt))
        pred_pose = torch.matmul(torch.from_numpy(jreg_t).to(dt)[None, :, :], pred_mesh * 1000)
        l1 = crit[0](pred_mesh, t['mesh'], t['mesh_valid'])
        l2 = 1e-1 * crit[1](pred_mesh, t['mesh'])
        l4 = 1e-3 * crit[3](pred_pose, t['reg_pose3d'], t['reg_pose3d_valid'])
        l5 = 1e-3 * crit[4](lift_pose, t['lift_pose3d'], t['lift_pose3d_valid'])
        l3 = 20 * crit[2](pred_mesh, t['mesh'])
        loss = l1 + l2 + l4 + l5 + l3
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in mdl.named_parameters()}
        parts = np.array([float(v) for v in (l1, l2, l3, l4, l5, loss)])
        return parts, grads, pred_mesh.detach(), lift_pose.detach()

    parts32, g32, mesh32, _ = step(model, torch.float32)
    m64 = model.double()
    for blk in m64.pose_lifter.blocks:
        blk.adj = blk.adj.double()
        blk.gcn.adj = blk.gcn.adj.double()
    hp = m64.pose_lifter.get_hop_path_encoding
    hp.edg_adj = hp.edg_adj.double()
    hp.spatial = hp.spatial.double()
    parts64, g64, mesh64, lift64 = step(m64, torch.float64)
    names = sorted(g64.keys())
    prs = np.random.RandomState(1234)
    out = dict(seed=np.int64(seed), alpha=np.bool_(alpha), num_joint=np.int64(J), batch=np.int64(B), pose2d=pose2d,
               loss_parts_f32=parts32, loss_parts_f64=parts64, param_names=np.array(names),
               mesh_f64_probe=mesh64.numpy()[:, ::97], lift_pose_f64=lift64.numpy())
    probe_idx, probe64, probe32, gmax, gnorm, noise = [], [], [], [], [], []
    for k in names:
        a64, a32 = g64[k].reshape(-1).numpy(), g32[k].reshape(-1).double().numpy()
        idx = np.sort(prs.choice(a64.size, size=min(48, a64.size), replace=False))
        pad = np.full(48, -1, np.int64)
        pad[:idx.size] = idx
        v64, v32 = np.zeros(48), np.zeros(48)
        v64[:idx.size], v32[:idx.size] = a64[idx], a32[idx]
        probe_idx.append(pad); probe64.append(v64); probe32.append(v32)
        gmax.append(np.abs(a64).max()); gnorm.append(np.sqrt((a64 ** 2).sum())); noise.append(np.abs(a32 - a64).max())
    out.update(probe_idx=np.stack(probe_idx), grad_f64=np.stack(probe64), grad_f32=np.stack(probe32), grad_absmax=np.array(gmax),
               grad_norm=np.array(gnorm), ref32_minus_f64_max=np.array(noise))
    np.savez(os.path.join(OUT, 'train_' + name + '.npz'), **out)
    rel = np.array(noise) / np.maximum(np.array(gmax), 1e-300)
    print('train_%s: loss %.6f (fp32 %.6f) parts %s; %d parameter tensors, ref fp32-vs-fp64 gradient error / max|g|: median %.1e max %.1e'
          % (name, parts64[-1], parts32[-1], parts64[:5].round(5).tolist(), len(names), np.median(rel), rel.max()))
    os.chdir(REPO)


def demo_preprocess_golden():
    """Config 1 plumbing (demo/run.py:193-198,124-133): raw COCO-17 joints -> [1,19,2] normalised input."""
    scratch = tempfile.mkdtemp(prefix='gator_golden_')
    install_shims(scratch, True)
    for m in [k for k in sys.modules if k in ('coord_utils', 'aug_utils')]:
        del sys.modules[m]
    import aug_utils
    import coord_utils
    raw = np.load(os.path.join(REF, 'demo', 'coco_joint_input.npy'))
    names = ('Nose', 'L_Eye', 'R_Eye', 'L_Ear', 'R_Ear', 'L_Shoulder', 'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist',
             'R_Wrist', 'L_Hip', 'R_Hip', 'L_Knee', 'R_Knee', 'L_Ankle', 'R_Ankle', 'Pelvis', 'Neck')
    j = raw.copy()

    def add(jc, a, b):
        n = (jc[names.index(a)] + jc[names.index(b)]) * 0.5
        n[2] = jc[names.index(a), 2] * jc[names.index(b), 2]
        return np.concatenate((jc, n.reshape(1, 3)))

    j = add(j, 'L_Hip', 'R_Hip')
    j = add(j, 'L_Shoulder', 'R_Shoulder')
    j = j[:, :2]
    bbox = coord_utils.get_bbox(j)
    bbox2 = coord_utils.process_bbox(bbox.copy())
    ji, _ = aug_utils.j2d_processing(j.copy(), (288, 384), bbox2, 0, 0, None)
    ji = ji[:, :2]
    ji /= np.array([[288, 384]])
    mean, std = np.mean(ji, axis=0), np.std(ji, axis=0)
    ji = (ji.copy() - mean) / std
    np.savez(os.path.join(OUT, 'demo_preprocess.npz'), raw_coco17=raw, pose2d=ji[None].astype(np.float32))
    print('demo: normalised rows 0-3', ji[:4].round(4).tolist(), 'min %.4f max %.4f' % (ji.min(), ji.max()))


COCO19_FLIP_PAIRS = ((1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))     # data/PW3D/dataset.py:33-34


def preprocess_chain_golden():
    """The GENERAL input chain run as it is (lib/coord_utils.py:21-66, lib/aug_utils.py:51-64,140-179; call order of
    data/PW3D/dataset.py:236-250): random 19-joint detections, rotation in +-30 degrees, random flips, plus boxes that
    process_bbox rejects (None)."""
    scratch = tempfile.mkdtemp(prefix='gator_golden_')
    install_shims(scratch, True)
    for m in [k for k in sys.modules if k in ('coord_utils', 'aug_utils')]:
        del sys.modules[m]
    import aug_utils
    import coord_utils
    rs = np.random.RandomState(777)
    N, J = 48, 19
    joints = np.zeros((N, J, 2), np.float64)
    rot = np.zeros(N)
    flip = np.zeros(N, np.int32)
    out = np.zeros((N, J, 2), np.float32)
    valid = np.ones(N, np.int32)
    for i in range(N):
        c = rs.rand(2) * np.array([1200, 800]) + 100
        ext = np.array([rs.rand() * 300 + 20, rs.rand() * 500 + 40]) * (0.3 + rs.rand())
        joints[i] = c + (rs.rand(J, 2) - 0.5) * ext
        if i % 3:
            rot[i] = np.clip(rs.randn() * 30, -60, 60)
        flip[i] = int(i % 4 == 1)
        if i in (7, 23):
            joints[i] = joints[i, :1]                       # all joints coincide: w = h = 0
        if i == 31:
            joints[i, :, 0] = joints[i, 0, 0] + np.linspace(0, 0.5, J)      # narrower than one pixel
        bbox = coord_utils.get_bbox(joints[i])
        bbox = coord_utils.process_bbox(bbox.copy())
        if bbox is None:
            valid[i] = 0
            continue
        ji, _ = aug_utils.j2d_processing(joints[i].copy(), (288, 384), bbox, rot[i], int(flip[i]), COCO19_FLIP_PAIRS)
        ji = ji[:, :2]
        ji /= np.array([[288, 384]])
        mean, std = np.mean(ji, axis=0), np.std(ji, axis=0)
        out[i] = (ji.copy() - mean) / std
    np.savez(os.path.join(OUT, 'preprocess_chain.npz'), joints=joints, rot=rot, flip=flip, flip_pairs=np.array(COCO19_FLIP_PAIRS, np.int32),
             pose2d=out, valid=valid)
    print('preprocess chain: %d samples, %d rejected by process_bbox' % (N, int((valid == 0).sum())))


def rigid_align_golden():
    """lib/coord_utils.py:127-149 run as it is: random, mirrored (det < 0 branch), near-coplanar and scaled point sets, plus the
    PA-MPJPE reduction of data/PW3D/dataset.py:337-375 (per-sample rigid_align on the 14 evaluation joints, mean distance)."""
    scratch = tempfile.mkdtemp(prefix='gator_golden_')
    install_shims(scratch, True)
    for m in [k for k in sys.modules if k in ('coord_utils',)]:
        del sys.modules[m]
    import coord_utils
    rs = np.random.RandomState(4242)
    A, B = [], []
    for i in range(24):
        a = rs.randn(14, 3) * 120
        q, _ = np.linalg.qr(rs.randn(3, 3))
        b = (0.5 + rs.rand()) * a @ q.T + rs.randn(3) * 40 + rs.randn(14, 3) * (8 if i % 2 else 0.0)
        if i % 4 == 1:
            b[:, 0] = -b[:, 0]                      # mirror image: forces the reflection fix
        if i % 6 == 2:
            a[:, 2] *= 1e-3                         # nearly coplanar: smallest singular value ~ 0
        A.append(a)
        B.append(b)
    A, B = np.stack(A), np.stack(B)
    out = np.stack([coord_utils.rigid_align(a, b) for a, b in zip(A, B)])
    pa = float(np.mean([np.sqrt(((o - b) ** 2).sum(1)).mean() for o, b in zip(out, B)]))
    A32, B32 = A.astype(np.float32), B.astype(np.float32)      # the reference feeds float32 arrays from the model
    out32 = np.stack([coord_utils.rigid_align(a, b) for a, b in zip(A32, B32)])
    np.savez(os.path.join(OUT, 'rigid_align.npz'), A=A, B=B, aligned=out, pa_mpjpe=np.float64(pa), aligned_from_f32=out32)
    print('rigid_align: 24 sets, PA error %.4f mm' % pa)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'base' in only:
        dump_jregressors()
        run_variant('h36m17_bn', 17, False, seed=0)
        run_variant('coco19_alpha', 19, True, seed=100)
        demo_preprocess_golden()
    if not only or 'scale' in only:
        sub = np.sort(np.random.RandomState(99).permutation(6890)[:512])
        run_variant('h36m17_bn', 17, False, seed=0, B=64, upsample_gain=0.2, subset=sub, out_name='scale_gain02')
        run_variant('h36m17_bn', 17, False, seed=0, B=64, upsample_gain=1.0, subset=sub, out_name='scale_gain10')
    if not only or 'train' in only:
        train_golden('h36m17_bn', 17, False, seed=0)
        train_golden('coco19_alpha', 19, True, seed=100)
    if not only or 'rigid' in only:
        rigid_align_golden()
    if not only or 'preprocess' in only:
        preprocess_chain_golden()
