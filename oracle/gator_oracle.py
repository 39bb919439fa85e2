"""Oracle (test infrastructure): CPU restatement of the GATOR eval forward.

Functional torch-CPU code, dtype-generic (float32 = the reference's arithmetic,
float64 = the tolerance anchor), fed only with a reference-layout ``state_dict``
and the base-data arrays.  It follows the reference's op order and uses the same
aten ops (F.linear, F.layer_norm, softmax, F.conv1d, ...) so that, timed on host
cores, it is a fair stand-in for the reference's CPU path (bench.py
``cpu_baseline.kind = "port"``).

Reference files (relative to /root/reference) are cited per function.  Pinned
against the real reference by tools/gen_golden.py -> tests/golden/*.npz.

NEVER imported by gator_amd/ (see oracle/__init__.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import graph_consts as gc

NUM_HEADS = 8        # lib/models/GAT.py:46 (num_heads default), lib/core/base.py:57
MDR_DIM = 64         # lib/models/MDR.py:74
MDR_HEADS = 2        # lib/models/MDR.py:96-97


class Consts:
    """Load-time constants the reference keeps as plain attributes (not in state_dict)."""

    def __init__(self, num_joint, j_regressor, base, alpha):
        """base: dict with smpl_mean_vertices [6890,3] f32, D (list of 2 scipy sparse), J_regressor_h36m
        [17,6890], optional shortest_path / path.  lib/models/GAT.py:56-112, lib/models/MDR.py:72-90."""
        self.J = int(num_joint)
        self.alpha = bool(alpha)
        mv = np.asarray(base['smpl_mean_vertices'], np.float32)
        g = gc.gat_constants(self.J, np.asarray(j_regressor, np.float32), mv,
                             base.get('shortest_path'), base.get('path'))
        self.graph_adj = g['graph_adj']
        self.sp = g['shortest_path']
        self.path = g['path']
        self.edge_input = g['edge_input']
        self.max_dist = g['max_dist']
        self.v6890 = mv
        self.v431 = gc.downsample(mv, base['D'])
        # MDR.py:85-87 -- ALWAYS the 17-joint h36m regressor, whatever the input joint set
        jt = np.asarray(base['J_regressor_h36m'], np.float32) @ mv
        self.vj = gc.build_verts_joints_relation(jt, self.v431)


def _t(x, dtype):
    return torch.as_tensor(np.asarray(x)).to(dtype) if not torch.is_tensor(x) else x.to(dtype)


def hop_path_bias(sd, c, dtype, p='pose_lifter.get_hop_path_encoding.'):
    """HopPathEncoding.forward, lib/models/backbones/modules.py:98-107 (+ ctor :81-96)."""
    J, H = c.J, NUM_HEADS
    sp = torch.as_tensor(c.sp).long()
    spatial = sp - 1
    spatial = torch.where(spatial > 0, spatial, torch.ones_like(sp))
    spatial = (1.0 / spatial.expand(H, -1, -1)).to(dtype)          # modules.py:88-93 (float32 in the reference)
    spb = F.embedding(sp, sd[p + 'spatial_pos_encoder.weight'].to(dtype), padding_idx=0).permute(2, 0, 1)   # padding_idx: row 0 gets no gradient
    ea = _t(c.edge_input, dtype).permute(2, 0, 1)
    ea = F.linear(ea.reshape(-1, J * J), sd[p + 'edge_encoder.weight'].to(dtype),
                  sd[p + 'edge_encoder.bias'].to(dtype)).reshape(-1, H, J, J)
    ea = ea.permute(1, 2, 3, 0)
    eb = (sd[p + 'W'].to(dtype) * ea).sum(-1)
    return spb + eb * spatial                                       # [H,J,J]


def gat_forward(sd, c, pose2d, dtype=torch.float32, taps=None, p='pose_lifter.'):
    """GAT.forward lib/models/GAT.py:133-152 with GATBlock :33-43 and the modules it calls
    (lib/models/backbones/modules.py: GraphLinear :49-50, Attention :121-138, MGCN :243-255,
    X_Feat :158-177, MLP :188-196).  pose2d [B,J,2] -> (x_out [B,3J] mm, feat [B,J,128])."""
    g = lambda k: sd[p + k].to(dtype)
    B, J, H = pose2d.shape[0], c.J, NUM_HEADS
    x = pose2d.to(dtype).reshape(B, J, 2).permute(0, 2, 1)                          # [B,2,J]
    x = torch.matmul(g('GLinear.0.W')[None], x) + g('GLinear.0.b')[None, :, None]   # GraphLinear
    x = F.group_norm(x, 4, g('GLinear.1.weight'), g('GLinear.1.bias'), 1e-5)
    x = F.gelu(x)
    x = torch.matmul(g('GLinear.3.W')[None], x) + g('GLinear.3.b')[None, :, None]
    x = x.permute(0, 2, 1)                                                          # [B,J,C]
    C = x.shape[-1]
    x = x + F.embedding(torch.arange(1, J + 1), g('pos_id_embed.weight'))
    deg = torch.as_tensor(c.graph_adj).long().sum(1).view(-1)                       # GAT.py:143
    x = x + F.embedding(deg, g('pos_num_embed.weight'))
    bias = hop_path_bias(sd, c, dtype, p + 'get_hop_path_encoding.')
    if taps is not None:
        taps['hop_path_bias'] = bias
        taps['gat_embed'] = x
    sp = torch.as_tensor(c.sp).long()
    m1 = (sp <= 1).to(dtype)
    m2 = (sp == 2).to(dtype)
    A = _t(c.graph_adj, dtype)
    E = torch.eye(J, dtype=dtype)
    for i in range(6):
        b = 'blocks.%d.' % i
        y = F.layer_norm(x, (C,), g(b + 'norm1.weight'), g(b + 'norm1.bias'), 1e-5)
        # Attention
        qkv = F.linear(y, g(b + 'attn.qkv.weight'), g(b + 'attn.qkv.bias'))
        qkv = qkv.reshape(B, J, 3, H, C // H).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q @ k.transpose(-2, -1)) * ((C // H) ** -0.5)
        att = att + bias.expand(B, -1, -1, -1)
        att = att.softmax(dim=-1)
        a = (att @ v).transpose(1, 2).reshape(B, J, C)
        a = F.linear(a, g(b + 'attn.proj.weight'), g(b + 'attn.proj.bias'))
        # MGCN
        W = g(b + 'gcn.W')
        h0 = torch.matmul(y, W[0])
        h1 = torch.matmul(y, W[1])
        adj = A + g(b + 'gcn.adj2')
        adj = (adj.T + adj) / 2
        M = g(b + 'gcn.M')
        gout = torch.matmul(adj * E, M * h0) + torch.matmul(adj * (1 - E), M * h1) + g(b + 'gcn.bias').view(1, 1, -1)
        s = a + gout
        # X_Feat
        f0 = torch.bmm(m1.expand(B, -1, -1), F.linear(s, g(b + 'x_feat.linears.0.weight'), g(b + 'x_feat.linears.0.bias')))
        f1 = torch.bmm(m2.expand(B, -1, -1), F.linear(s, g(b + 'x_feat.linears.1.weight'), g(b + 'x_feat.linears.1.bias')))
        xf = F.linear(torch.cat([f0, f1], -1), g(b + 'x_feat.linearback.weight'), g(b + 'x_feat.linearback.bias'))
        x = x + xf
        # MLP
        y2 = F.layer_norm(x, (C,), g(b + 'norm2.weight'), g(b + 'norm2.bias'), 1e-5)
        hdn = F.gelu(F.linear(y2, g(b + 'mlp.fc1.weight'), g(b + 'mlp.fc1.bias')))
        x = x + F.linear(hdn, g(b + 'mlp.fc2.weight'), g(b + 'mlp.fc2.bias'))
        if taps is not None:
            taps['gat_block%d' % i] = x
    x = F.layer_norm(x, (C,), g('norm.weight'), g('norm.bias'), 1e-5)
    feat = F.gelu(x)
    x_out = F.linear(feat.reshape(B, -1), g('lifter.weight'), g('lifter.bias'))
    return x_out, feat


_ATTN_KEPT = {}
KEEP_ATTENTION_MAPS = False     # timing runs only (bench.py cpu_baseline, tools/cpu_port_vs_reference.py): the reference's MultiHeadedAttention keeps self.attn
                                # (vanilla_transformer_encoder.py:91), so its three maps stay allocated between forwards and the next forward's temporaries land on
                                # warm pages; a port that frees them pays first-touch page faults the reference does not (0.73 against 0.93 of its speed)


def _custom_ln(x, a2, b2, eps=1e-6):
    """lib/models/vanilla_transformer_encoder.py:31-34 -- unbiased std, eps added to std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a2 * (x - mean) / (std + eps) + b2


def mdr_forward(sd, c, pc, dtype=torch.float32, taps=None, p='pose2mesh.', train=False):
    """MDR.forward lib/models/MDR.py:124-170 (CrossAttentionBlock :64-69, CrossAttention :34-46,
    MultiHeadedAttention lib/models/vanilla_transformer_encoder.py:82-94, attention :36-46).
    pc [B,J,2+3+128] -> vertices [B,6890,3] (metres)."""
    g = lambda k: sd[p + k].to(dtype)
    B, J = pc.shape[0], c.J
    pc = pc.to(dtype)
    E, Hh, V = MDR_DIM, MDR_HEADS, c.v431.shape[0]
    v431 = _t(c.v431, dtype)
    vj = torch.as_tensor(c.vj).long()
    vf = torch.cat([v431.unsqueeze(0).expand(B, -1, -1), pc[:, vj, 2:5]], dim=2)
    jf = F.linear(pc, g('get_joint_feature.weight'), g('get_joint_feature.bias'))
    vf = F.linear(vf, g('get_verts_feature.weight'), g('get_verts_feature.bias'))
    jf = jf + F.embedding(torch.arange(1, J + 1), g('pos_j_id_embed.weight'))
    vf = vf + F.embedding(torch.arange(1, V + 1), g('pos_v_id_embed.weight'))
    if taps is not None:
        taps['mdr_vf0'] = vf
        taps['mdr_jf'] = jf
    d = E // Hh
    for li, sfx in enumerate(('', '_1', '_2')):
        e = 'encoder%s.' % sfx
        fz = F.layer_norm(torch.cat([vf, jf], 1), (E,), g(e + 'norm1.weight'), g(e + 'norm1.bias'), 1e-5)
        q = F.linear(fz[:, :V], g(e + 'attn.wq.weight')).reshape(B, V, Hh, d).permute(0, 2, 1, 3)
        k = F.linear(fz[:, V:], g(e + 'attn.wk.weight')).reshape(B, J, Hh, d).permute(0, 2, 1, 3)
        v = F.linear(fz[:, V:], g(e + 'attn.wv.weight')).reshape(B, J, Hh, d).permute(0, 2, 1, 3)
        att = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, V, E)
        vf = vf + F.linear(o, g(e + 'attn.proj.weight'), g(e + 'attn.proj.bias'))
        y = F.layer_norm(vf, (E,), g(e + 'norm2.weight'), g(e + 'norm2.bias'), 1e-5)
        vf = vf + F.linear(F.gelu(F.linear(y, g(e + 'mlp.fc1.weight'), g(e + 'mlp.fc1.bias'))),
                           g(e + 'mlp.fc2.weight'), g(e + 'mlp.fc2.bias'))
        if taps is not None:
            taps['mdr_cross%d' % li] = vf
        vf = _custom_ln(vf, g('norm%s.a_2' % sfx), g('norm%s.b_2' % sfx))
        sa = 'selfatt%s.linears.' % sfx
        qq, kk, vv = [F.linear(vf, g(sa + '%d.weight' % n), g(sa + '%d.bias' % n)).view(B, -1, Hh, d).transpose(1, 2)
                      for n in range(3)]
        sc = torch.matmul(qq, kk.transpose(-2, -1)) / math.sqrt(d)
        pa = F.softmax(sc, dim=-1)
        if KEEP_ATTENTION_MAPS:
            _ATTN_KEPT[li] = pa
        xo = torch.matmul(pa, vv).transpose(1, 2).contiguous().view(B, -1, E)
        vf = vf + F.linear(xo, g(sa + '3.weight'), g(sa + '3.bias'))
        if taps is not None:
            taps['mdr_lbf%d' % li] = vf
    ac = F.linear(vf, g('motion_linear.weight'), g('motion_linear.bias'))
    mat_a, mat_c = ac[:, :, :20], ac[:, :, -3:]
    mat_b = F.linear(vf, g('bias_linear.weight'), g('bias_linear.bias'))
    if c.alpha:
        mat_b = F.layer_norm(mat_b, (3,), g('bias_norm.weight'), g('bias_norm.bias'), 1e-5)
    elif train:  # .train(): batch statistics (running stats updated on copies; they do not enter the output)
        mat_b = F.batch_norm(mat_b, g('bias_norm.running_mean').clone(), g('bias_norm.running_var').clone(),
                             g('bias_norm.weight'), g('bias_norm.bias'), True, 0.1, 1e-5)
    else:  # BatchNorm1d(431) in eval mode: channels = the VERTEX axis (MDR.py:119,159)
        mat_b = F.batch_norm(mat_b, g('bias_norm.running_mean'), g('bias_norm.running_var'),
                             g('bias_norm.weight'), g('bias_norm.bias'), False, 0.1, 1e-5)
    mat_b = F.gelu(mat_b)
    mat_b = F.conv1d(mat_b, g('bias_conv1d.weight'), g('bias_conv1d.bias'), padding=1)
    if c.alpha:
        al = 1.1 ** F.linear(vf, g('scale_linear.weight'), g('scale_linear.bias'))
    else:
        al = 1
    vc = al * mat_a.softmax(dim=-1).bmm(mat_b) + mat_c
    if taps is not None:
        taps['vert431'] = vc
    out = F.conv1d(vc, g('upsample_conv.weight'), g('upsample_conv.bias'), padding=1)
    return out + _t(c.v6890, dtype)


def gator_forward(sd, c, pose2d, dtype=torch.float32, taps=None):
    """GATOR.forward lib/models/GATOR.py:16-22.  Returns (cam_mesh [B,6890,3] m, pose3d [B,J,3] mm)."""
    with torch.no_grad():
        pose2d = pose2d.to(dtype)
        x_out, feat = gat_forward(sd, c, pose2d.reshape(len(pose2d), -1), dtype, taps)
        pose3d = x_out.reshape(-1, c.J, 3)
        pc = torch.cat((pose2d, pose3d / 1000, feat), dim=2)
        if taps is not None:
            taps['feat'] = feat
            taps['pose3d'] = pose3d
        mesh = mdr_forward(sd, c, pc, dtype, taps)
    return mesh, pose3d


# ---- training row: the differentiable forward and the losses, for torch-CPU autograd as the gradient oracle ----

def gator_forward_train(sd, c, pose2d, dtype=torch.float64):
    """GATOR.forward in .train() mode with every dropout / DropPath probability at 0 (the reference's RNG stream is not
    reproducible elsewhere): differs from eval only in BatchNorm1d using batch statistics.  Differentiable w.r.t. sd tensors."""
    pose2d = pose2d.to(dtype)
    x_out, feat = gat_forward(sd, c, pose2d.reshape(len(pose2d), -1), dtype)
    pose3d = x_out.reshape(-1, c.J, 3)
    pc = torch.cat((pose2d, pose3d / 1000, feat), dim=2)
    return mdr_forward(sd, c, pc, dtype, train=True), pose3d


def coord_loss(pred, target, valid):
    """CoordLoss(has_valid=True), lib/core/loss.py:10-25"""
    return F.l1_loss(pred * valid, target * valid, reduction='mean')


def normal_vector_loss(coord_out, coord_gt, face):
    """NormalVectorLoss.forward, lib/core/loss.py:59-86"""
    face = torch.as_tensor(np.asarray(face)).long()
    nz = lambda v: F.normalize(v, p=2, dim=2)
    v1_out = nz(coord_out[:, face[:, 1], :] - coord_out[:, face[:, 0], :])
    v2_out = nz(coord_out[:, face[:, 2], :] - coord_out[:, face[:, 0], :])
    v3_out = nz(coord_out[:, face[:, 2], :] - coord_out[:, face[:, 1], :])
    v1_gt = nz(coord_gt[:, face[:, 1], :] - coord_gt[:, face[:, 0], :])
    v2_gt = nz(coord_gt[:, face[:, 2], :] - coord_gt[:, face[:, 0], :])
    normal_gt = nz(torch.cross(v1_gt, v2_gt, dim=2))
    cos = [torch.abs(torch.sum(v * normal_gt, 2, keepdim=True)) for v in (v1_out, v2_out, v3_out)]
    return torch.cat(cos, 1).mean()


def edge_length_loss(coord_out, coord_gt, face):
    """EdgeLengthLoss.forward, lib/core/loss.py:89-112"""
    face = torch.as_tensor(np.asarray(face)).long()
    def d(x, a, b):
        return torch.sqrt(torch.sum((x[:, face[:, a], :] - x[:, face[:, b], :]) ** 2, 2, keepdim=True))
    diffs = [torch.abs(d(coord_out, a, b) - d(coord_gt, a, b)) for a, b in ((0, 1), (0, 2), (1, 2))]
    return torch.cat(diffs, 1).mean()


def training_loss(pred_mesh, lift_pose, targets, j_regressor, face, with_edge=False, normal_weight=1e-1, edge_weight=20.0, joint_weight=1e-3):
    """Trainer.train's loss, lib/core/base.py:136-148 (weights: lib/core/config.py:58-60)."""
    dt = pred_mesh.dtype
    t = lambda k: targets[k].to(dt)
    pred_pose = torch.matmul(torch.as_tensor(np.asarray(j_regressor)).to(dt)[None], pred_mesh * 1000)
    parts = {
        'vertice': coord_loss(pred_mesh, t('mesh'), t('mesh_valid')),
        'normal': normal_weight * normal_vector_loss(pred_mesh, t('mesh'), face),
        'mesh2joint3d': joint_weight * coord_loss(pred_pose, t('reg_pose3d'), t('reg_pose3d_valid')),
        'liftedjoint3d': joint_weight * coord_loss(lift_pose, t('lift_pose3d'), t('lift_pose3d_valid')),
    }
    loss = parts['vertice'] + parts['normal'] + parts['mesh2joint3d'] + parts['liftedjoint3d']
    if with_edge:
        parts['edge'] = edge_weight * edge_length_loss(pred_mesh, t('mesh'), face)
        loss = loss + parts['edge']
    return loss, parts


# ---- caller-side pieces either side of the path ("next" rows) --------------------

def normalise_pose2d(joint_img):
    """Input contract a0: data/PW3D/dataset.py:244-250, demo/run.py:130-133 with rot=0/flip=0, where the
    bbox->affine->/[288,384] chain is a per-axis positive scale+shift that cancels in the standardisation."""
    j = np.asarray(joint_img, np.float64)[:, :2]
    return (j - j.mean(0)) / j.std(0)


def add_pelvis_neck_coco(joint17):
    """demo/run.py:103-121 (z := product of the two z's, dropped afterwards)."""
    j = np.asarray(joint17, np.float64)
    pel = (j[11] + j[12]) * 0.5
    pel[2] = j[11, 2] * j[12, 2]
    nk = (j[5] + j[6]) * 0.5
    nk[2] = j[5, 2] * j[6, 2]
    return np.concatenate([j, pel[None], nk[None]], 0)


def regress_joints(j_regressor, mesh):
    """lib/core/base.py:221, demo/run.py:142."""
    return torch.matmul(torch.as_tensor(j_regressor).to(mesh.dtype)[None], mesh)


def mpjpe(pred_joint, target_joint, eval_joints=None, root=0):
    """data/PW3D/dataset.py:273-286 (joint part): root-align then mean Euclidean distance."""
    p = pred_joint - pred_joint[:, root:root + 1]
    t = target_joint - target_joint[:, root:root + 1]
    if eval_joints is not None:
        p, t = p[:, eval_joints], t[:, eval_joints]
    return float(np.sqrt(((np.asarray(p) - np.asarray(t)) ** 2).sum(2)).mean())


def rigid_transform_3d(A, B):
    """lib/coord_utils.py:127-142 (numpy, float64)."""
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    n = A.shape[0]
    ca, cb = A.mean(0), B.mean(0)
    H = (A - ca).T @ (B - cb) / n
    U, s, V = np.linalg.svd(H)
    R = V.T @ U.T
    if np.linalg.det(R) < 0:
        s[-1] = -s[-1]
        V[2] = -V[2]
        R = V.T @ U.T
    c = 1 / np.var(A, axis=0).sum() * s.sum()
    t = -(c * R) @ ca + cb
    return c, R, t


def rigid_align(A, B):
    """lib/coord_utils.py:145-149."""
    c, R, t = rigid_transform_3d(A, B)
    return ((c * R) @ np.asarray(A, np.float64).T).T + t


def pa_mpjpe(pred_joint, target_joint, eval_joints=None):
    """data/PW3D/dataset.py:337-375 style: per-sample Procrustes alignment, then mean joint error."""
    errs = []
    for p, t in zip(np.asarray(pred_joint, np.float64), np.asarray(target_joint, np.float64)):
        if eval_joints is not None:
            p, t = p[list(eval_joints)], t[list(eval_joints)]
        errs.append(np.sqrt(((rigid_align(p, t) - t) ** 2).sum(1)).mean())
    return float(np.mean(errs))


# ---- the general input chain: bbox -> affine (rotation, flip) -> /[W,H] -> standardise ---------------------------------------
def get_bbox(joint_img):
    """lib/coord_utils.py:21-39 (tight box of the joints, float32)."""
    x, y = joint_img[:, 0], joint_img[:, 1]
    xmin, ymin, xmax, ymax = min(x), min(y), max(x), max(y)
    xc, w = (xmin + xmax) / 2., xmax - xmin
    xmin, xmax = xc - 0.5 * w, xc + 0.5 * w
    yc, h = (ymin + ymax) / 2., ymax - ymin
    ymin, ymax = yc - 0.5 * h, yc + 0.5 * h
    return np.array([xmin, ymin, xmax - xmin, ymax - ymin]).astype(np.float32)


def process_bbox(bbox, aspect_ratio):
    """lib/coord_utils.py:42-66: None for a degenerate box (width or height under one pixel), else the aspect-preserving box."""
    x, y, w, h = bbox
    x1, y1, x2, y2 = x, y, x + (w - 1), y + (h - 1)
    if w * h > 0 and x2 >= x1 and y2 >= y1:
        bbox = np.array([x1, y1, x2 - x1, y2 - y1])
    else:
        return None
    w, h = bbox[2], bbox[3]
    cx, cy = bbox[0] + w / 2., bbox[1] + h / 2.
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    bbox[2], bbox[3] = w, h
    bbox[0], bbox[1] = cx - bbox[2] / 2., cy - bbox[3] / 2.
    return bbox


def affine_from_bbox(bbox, rot, res):
    """get_center_scale (lib/coord_utils.py:7-18) + get_affine_transform (lib/aug_utils.py:140-173) with shift = 0: three point
    pairs in float32, solved for the 2x3 matrix as cv2.getAffineTransform does (float64)."""
    x, y, w, h = bbox
    center = np.array([x + w * 0.5, y + h * 0.5], np.float32)
    scale = np.array([w, h], np.float32)
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = [0, scale[0] * -0.5]
    src_dir = [p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs]                  # get_dir, aug_utils.py:188-195
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center + src_dir
    dst[0] = [res[0] * 0.5, res[1] * 0.5]
    dst[1] = np.array([res[0] * 0.5, res[1] * 0.5]) + np.array([0, res[0] * -0.5], np.float32)
    for m in (src, dst):                                                      # get_3rd_point, aug_utils.py:182-184
        d = m[0] - m[1]
        m[2] = m[1] + np.array([-d[1], d[0]], np.float32)
    a = np.concatenate([src.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(a, dst.astype(np.float64)).T                       # [2,3]


def preprocess_pose2d(joint_img, rot=0.0, flip=False, flip_pairs=(), res=(288, 384)):
    """data/PW3D/dataset.py:236-250 / demo/run.py:124-134: bbox -> process_bbox -> j2d_processing (lib/aug_utils.py:51-64:
    affine per joint, optional flip_2d_joint :31-38, float32 cast) -> /[W,H] -> per-axis standardisation.  None if the box is
    degenerate (the datasets drop such samples)."""
    kp = np.array(joint_img, dtype=np.float64)[:, :2].copy()
    bbox = process_bbox(get_bbox(kp).copy(), res[0] / res[1])
    if bbox is None:
        return None
    t = affine_from_bbox(bbox, rot, res)
    for i in range(kp.shape[0]):
        kp[i] = t @ np.array([kp[i, 0], kp[i, 1], 1.])
    if flip:
        kp[:, 0] = res[0] - kp[:, 0] - 1
        for a, b in flip_pairs:
            kp[a], kp[b] = kp[b].copy(), kp[a].copy()
    kp = kp.astype(np.float32)
    kp /= np.array([[res[0], res[1]]])
    mean, std = np.mean(kp, axis=0), np.std(kp, axis=0)
    return (kp.copy() - mean) / std
