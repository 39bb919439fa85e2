"""GATOR in TRAINING mode on the library's differentiable primitives (gator_amd/train/ops.py): the same arithmetic as the
reference modules in ``.train()`` - dropout / DropPath at the reference's sites and rates, BatchNorm1d on batch statistics -
written functionally over a reference-layout parameter dict, so a reference checkpoint trains here unchanged.

Reference: lib/models/GATOR.py:16-22, GAT.py:33-43,133-152, backbones/modules.py:49-50,98-107,121-138,158-177,188-196,243-255,
MDR.py:34-46,64-69,124-170, vanilla_transformer_encoder.py:31-46,82-94; timm DropPath / Mlp semantics."""
import math

import numpy as np
import torch

from . import ops

NUM_HEADS, EMBED, DEPTH = 8, 128, 6          # lib/core/base.py:57, lib/models/GAT.py:46
MDR_DIM, MDR_HEADS, NV431 = 64, 2, 431       # lib/models/MDR.py:74,96-97,80-81
BUFFER_SUFFIXES = ('graph_adj', 'init_vertices', 'init_vertices_6890', 'running_mean', 'running_var', 'num_batches_tracked')


def is_buffer(key):
    """state_dict entries that are buffers in the reference (register_buffer / BatchNorm statistics), not parameters."""
    return key.rsplit('.', 1)[-1] in BUFFER_SUFFIXES


class Rates:
    """Dropout / DropPath probabilities at the reference's defaults (lib/models/GAT.py:47,114-119, modules.py:189, MDR.py:49-50,76,97)."""

    def __init__(self, scale=1.0):
        self.gat_attn = 0.4 * scale
        self.gat_proj = 0.4 * scale
        self.gat_mlp = 0.1 * scale
        self.gat_path = [0.2 * scale * i / (DEPTH - 1) for i in range(DEPTH)]      # torch.linspace(0, 0.2, depth)
        self.mdr_attn = 0.2 * scale
        self.mdr_drop = 0.2 * scale
        self.mdr_path = 0.2 * scale
        self.mdr_self = 0.1 * scale


class Consts:
    """Input-independent device constants of one model (the reference keeps them as plain attributes)."""

    def __init__(self, J, graph_adj, sp, edge_input, v431, v6890, vj, alpha, device):
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(device)
        self.J, self.alpha = int(J), bool(alpha)
        sp = np.asarray(sp).astype(np.int64)
        self.A = f(graph_adj)
        self.E = f(np.eye(J))
        self.notE = f(1.0 - np.eye(J))
        self.m1 = f(sp <= 1)                                                  # modules.py:163-170
        self.m2 = f(sp == 2)
        deg = np.asarray(graph_adj).astype(np.int64).sum(1)                   # GAT.py:143
        self.deg_onehot = f(np.eye(J)[deg])                                   # [J, J] rows of pos_num_embed
        ns = 10                                                               # num_spatial, GAT.py:111
        oh = np.eye(ns)[sp.reshape(-1)]                                       # [J*J, 10]
        self.sp_onehot_pad = f(oh * (np.arange(ns) == 0))                     # padding_idx = 0: used in the forward, no gradient
        self.sp_onehot = f(oh * (np.arange(ns) != 0))
        spatial = np.where(sp - 1 > 0, sp - 1, 1)
        self.inv_spatial = f(1.0 / spatial)                                   # modules.py:88-93
        ei = np.asarray(edge_input, np.float32)                               # [J,J,D]
        self.D = ei.shape[2]
        self.edge_rows = f(ei.transpose(2, 0, 1).reshape(self.D, J * J))
        self.v431 = f(v431)
        self.v6890 = f(v6890)
        self.vj_onehot = f(np.eye(J)[np.asarray(vj).astype(np.int64)])        # [431, J]: pc[:, vj] as a product


def consts_from_module(m, device):
    """Constants of a gator_amd.models.GATOR module (the drop-in built by models.GATOR.get_model): the same tables its
    inference context is created from."""
    sd = m.state_dict()
    return Consts(m.num_joint, sd['pose_lifter.graph_adj'].cpu().numpy(), m.pose_lifter.spatial_pos, m.pose_lifter.edge_input,
                  sd['pose2mesh.init_vertices'].cpu().numpy(), sd['pose2mesh.init_vertices_6890'].cpu().numpy(), m.pose2mesh.vj_relation,
                  m.pose2mesh.alpha, device)


class _Unfold3(torch.autograd.Function):
    """x [B,C,3] -> U [B,3,C,3], U[b,t,c,k] = x[b,c,t+k-1] (0 outside): the im2col of a kernel-3 / padding-1 Conv1d over the
    xyz axis, so that the convolution is ONE GEMM with the weight viewed as [O, 3C] (MDR.py:121-122)."""

    @staticmethod
    def forward(ctx, x):
        B, C, _ = x.shape
        u = ops.zeros((B, 3, C, 3), x.device)
        for k in range(3):
            t0, t1 = max(0, 1 - k), min(3, 4 - k)
            ops.raw_unary(ops.U_AFFINE, x.narrow(2, t0 + k - 1, t1 - t0).transpose(1, 2), 1.0, 0.0, out=u.narrow(1, t0, t1 - t0).select(3, k))
        return u

    @staticmethod
    def backward(ctx, g):
        B, _, C, _ = g.shape
        dx = ops.zeros((B, C, 3), g.device)
        for k in range(3):
            t0, t1 = max(0, 1 - k), min(3, 4 - k)
            dst = dx.narrow(2, t0 + k - 1, t1 - t0)
            ops.raw_binary(ops.ADD, dst, g.narrow(1, t0, t1 - t0).select(3, k).transpose(1, 2), out=dst)
        return dx


def conv1d_k3(x, w, b):
    """F.conv1d(x [B,C,3], w [O,C,3], b, padding=1) -> [B,O,3]"""
    B, C, _ = x.shape
    u = _Unfold3.apply(x).reshape(B, 3, C * 3)
    wr = w.reshape(w.shape[0], C * 3)
    slot = getattr(w, '_gslot', None)                                # (forwarded, not consumed: ops.linear below is the one writer)
    if slot is not None:
        wr._gslot = slot.reshape(w.shape[0], C * 3)                  # the view keeps its slice of the flat gradient buffer
    y = ops.linear(u, wr, b)                                         # [B,3,O]
    return y.transpose(1, 2)


def hop_path_bias(P, c, p='pose_lifter.get_hop_path_encoding.'):
    """HopPathEncoding.forward, lib/models/backbones/modules.py:98-107 -> [H,J,J]"""
    J, H = c.J, NUM_HEADS
    w0, w1 = ops.fork(P[p + 'spatial_pos_encoder.weight'])
    spb = ops.add(ops.matmul(c.sp_onehot, w0), ops.matmul(c.sp_onehot_pad, w1.detach()))       # [J*J, H]
    spb = spb.reshape(J, J, H).permute(2, 0, 1)
    ea = ops.linear(c.edge_rows, P[p + 'edge_encoder.weight'], P[p + 'edge_encoder.bias'])     # [D, J*J*H]
    ea = ea.reshape(c.D, H, J, J).permute(1, 2, 3, 0)                                            # [H,J,J,D]
    eb = ops.sum_(ops.mul(P[p + 'W'], ea), [3])
    return ops.add(spb, ops.mul(eb, c.inv_spatial))


import os

# GATOR_TRAIN_FUSED_ATTN=0: attention cores and the MGCN aggregation composed from the primitives (the cross-check form) instead of
# their one-launch kernels
FUSED_SELF_ATTENTION = os.environ.get('GATOR_TRAIN_FUSED_ATTN', '1') != '0'


def gat_block(P, c, x, bias, i, gen, rates, training=True, p='pose_lifter.'):
    """GATBlock.forward (lib/models/GAT.py:33-43) composed from the primitives: x [B,J,128] -> [B,J,128]."""
    g = lambda k: P[p + k]
    B, J, H, C = x.shape[0], c.J, NUM_HEADS, EMBED
    scale = (C // H) ** -0.5
    b = 'blocks.%d.' % i
    y, res = ops.layernorm_skip(x, g(b + 'norm1.weight'), g(b + 'norm1.bias'), 1e-5, 0)
    y, y0, y1 = ops.fork(y, 3)
    # Attention (modules.py:121-138)
    qkv = ops.linear(y, g(b + 'attn.qkv.weight'), g(b + 'attn.qkv.bias'))
    if FUSED_SELF_ATTENTION:            # one launch per direction, one wave per (sample, head) (csrc/train_attn.inc)
        a = ops.attention_small(qkv, bias, H, scale, rates.gat_attn, gen, training)
    else:
        q, k, v = [ops.reshape(t, B, J, H, C // H).permute(0, 2, 1, 3) for t in ops.split(qkv.reshape(B, J, 3, C), 2, (1, 1, 1))]
        att = ops.add(ops.matmul(q, k.transpose(-2, -1), scale), bias)
        att = ops.dropout(ops.softmax(att), rates.gat_attn, gen, training)
        a = ops.contiguous(ops.matmul(att, v).transpose(1, 2)).reshape(B, J, C)
    a = ops.dropout(ops.linear(a, g(b + 'attn.proj.weight'), g(b + 'attn.proj.bias')), rates.gat_proj, gen, training)
    # MGCN (modules.py:243-255)
    h0 = ops.xw(y0, g(b + 'gcn.W'), 0)
    h1 = ops.xw(y1, g(b + 'gcn.W'), 1)
    if FUSED_SELF_ATTENTION:            # diag / off-diagonal aggregation, modulation and bias in one launch per direction
        gout = ops.mgcn(h0, h1, sym_adjacency(c, g(b + 'gcn.adj2')), g(b + 'gcn.M'), g(b + 'gcn.bias'))
    else:
        adj_d, adj_o = ops.fork(sym_adjacency(c, g(b + 'gcn.adj2')))
        M0, M1 = ops.fork(g(b + 'gcn.M'))
        gout = ops.add(ops.add(ops.matmul(ops.mul(adj_d, c.E), ops.mul(M0, h0)), ops.matmul(ops.mul(adj_o, c.notE), ops.mul(M1, h1))),
                       g(b + 'gcn.bias').reshape(1, 1, -1))
    s = ops.drop_path(ops.add(a, gout), rates.gat_path[i], gen, training)
    # X_Feat (modules.py:158-177)
    s0, s1 = ops.fork(s)
    l0, l1 = ops.linear_group([(s0, g(b + 'x_feat.linears.0.weight'), g(b + 'x_feat.linears.0.bias')),
                               (s1, g(b + 'x_feat.linears.1.weight'), g(b + 'x_feat.linears.1.bias'))])
    f0, f1 = ops.matmul(c.m1, l0), ops.matmul(c.m2, l1)
    xf = ops.linear(ops.cat([f0, f1], 2), g(b + 'x_feat.linearback.weight'), g(b + 'x_feat.linearback.bias'))
    x = ops.add(res, xf)
    # MLP (modules.py:188-196)
    y2, res = ops.layernorm_skip(x, g(b + 'norm2.weight'), g(b + 'norm2.bias'), 1e-5, 0)
    if FUSED_SELF_ATTENTION:            # GELU + dropout, and dropout + DropPath + residual, one launch each
        hdn = ops.drop_fused(ops.linear(y2, g(b + 'mlp.fc1.weight'), g(b + 'mlp.fc1.bias')), None, True, rates.gat_mlp, 0.0, gen, training)
        return ops.drop_fused(ops.linear(hdn, g(b + 'mlp.fc2.weight'), g(b + 'mlp.fc2.bias')), res, False, rates.gat_mlp, rates.gat_path[i], gen, training)
    hdn = ops.dropout(ops.gelu(ops.linear(y2, g(b + 'mlp.fc1.weight'), g(b + 'mlp.fc1.bias'))), rates.gat_mlp, gen, training)
    m = ops.dropout(ops.linear(hdn, g(b + 'mlp.fc2.weight'), g(b + 'mlp.fc2.bias')), rates.gat_mlp, gen, training)
    return ops.add(res, ops.drop_path(m, rates.gat_path[i], gen, training))


def sym_adjacency(c, adj2):
    """MGCN's adjacency (modules.py:247-248): ((A + adj2)^T + (A + adj2)) / 2"""
    adj_a, adj_b = ops.fork(ops.add(c.A, adj2))
    return ops.affine(ops.add(adj_a.t(), adj_b), 0.5)


def gat_forward(P, c, pose2d, gen, rates, training=True, p='pose_lifter.'):
    """GAT.forward (lib/models/GAT.py:133-152) in training mode.  pose2d [B,J,2] -> (x_out [B,3J] mm, feat [B,J,128])."""
    g = lambda k: P[p + k]
    B, J, H, C = pose2d.shape[0], c.J, NUM_HEADS, EMBED
    x = pose2d.reshape(B, J, 2).permute(0, 2, 1)                                                # [B,2,J]
    x = ops.add(ops.matmul(g('GLinear.0.W'), x), g('GLinear.0.b').reshape(1, -1, 1))            # GraphLinear, modules.py:49-50
    x = ops.layernorm(x.reshape(B, 4, 16 * J), None, None, 1e-5, 0).reshape(B, 64, J)           # GroupNorm(4, 64)
    x = ops.add(ops.mul(x, g('GLinear.1.weight').reshape(1, -1, 1)), g('GLinear.1.bias').reshape(1, -1, 1))
    x = ops.gelu(x)
    x = ops.add(ops.matmul(g('GLinear.3.W'), x), g('GLinear.3.b').reshape(1, -1, 1))
    x = x.permute(0, 2, 1)                                                                      # [B,J,C]
    x = ops.add(x, ops.narrow(g('pos_id_embed.weight'), 0, 1, J))
    x = ops.add(x, ops.matmul(c.deg_onehot, g('pos_num_embed.weight')))
    biases = ops.fork(hop_path_bias(P, c, p + 'get_hop_path_encoding.'), DEPTH)
    for i in range(DEPTH):
        x = gat_block(P, c, x, biases[i], i, gen, rates, training, p)
    feat = ops.gelu(ops.layernorm(x, g('norm.weight'), g('norm.bias'), 1e-5, 0))
    feat, f2 = ops.fork(feat)
    x_out = ops.linear(f2.reshape(B, J * C), g('lifter.weight'), g('lifter.bias'))
    return x_out, feat


def _batchnorm_train(x, w, b, run_mean, run_var, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d(431) in training mode on x [B,431,3] (channels = the vertex axis, MDR.py:119,159): batch statistics over
    (B, xyz), biased variance for the normalisation, running statistics updated with the unbiased one."""
    n = x.shape[0] * x.shape[2]
    x, x2 = ops.fork(x)
    mu = ops.mean(x2, [0, 2], keepdim=True)
    mu, mu_s = ops.fork(mu)
    xc = ops.sub(x, mu)
    xc, xc2 = ops.fork(xc)
    var = ops.mean(ops.square(xc2), [0, 2], keepdim=True)
    var, var_s = ops.fork(var)
    y = ops.mul(xc, ops.rsqrt(ops.affine(var, 1.0, eps)))
    if run_mean is not None:                                  # running = (1-m) running + m batch  (no gradient)
        rm, rv = run_mean.reshape(1, -1, 1), run_var.reshape(1, -1, 1)
        ops.raw_binary(ops.ADD, ops.raw_unary(ops.U_AFFINE, rm, 1.0 - momentum, 0.0), ops.raw_unary(ops.U_AFFINE, mu_s.detach(), momentum, 0.0), out=rm)
        ops.raw_binary(ops.ADD, ops.raw_unary(ops.U_AFFINE, rv, 1.0 - momentum, 0.0),
                       ops.raw_unary(ops.U_AFFINE, var_s.detach(), momentum * n / max(n - 1, 1), 0.0), out=rv)
    return ops.add(ops.mul(y, w.reshape(1, -1, 1)), b.reshape(1, -1, 1))


def mdr_forward(P, c, pc, gen, rates, training=True, buffers=None, p='pose2mesh.'):
    """MDR.forward (lib/models/MDR.py:124-170) in training mode.  pc [B,J,2+3+128] -> vertices [B,6890,3] (metres)."""
    g = lambda k: P[p + k]
    B, J = pc.shape[0], c.J
    E, Hh, V = MDR_DIM, MDR_HEADS, NV431
    d = E // Hh
    pc, pc2 = ops.fork(pc)
    near = ops.matmul(c.vj_onehot, ops.narrow(pc2, 2, 2, 3))                                    # pc[:, vj, 2:5]
    vf = ops.cat([c.v431.unsqueeze(0).expand(B, -1, -1), near], 2)
    jf = ops.linear(pc, g('get_joint_feature.weight'), g('get_joint_feature.bias'))
    vf = ops.linear(vf, g('get_verts_feature.weight'), g('get_verts_feature.bias'))
    jf = ops.add(jf, ops.narrow(g('pos_j_id_embed.weight'), 0, 1, J))
    vf = ops.add(vf, ops.narrow(g('pos_v_id_embed.weight'), 0, 1, V))
    jfs = ops.fork(jf, 3)
    for li, sfx in enumerate(('', '_1', '_2')):
        e = 'encoder%s.' % sfx
        vf, res = ops.fork(vf)
        fz = ops.layernorm(ops.cat([vf, jfs[li]], 1), g(e + 'norm1.weight'), g(e + 'norm1.bias'), 1e-5, 0)
        fq, fj = ops.split(fz, 1, (V, J))
        fk, fv = ops.fork(ops.contiguous(fj))
        q, k, v = ops.linear_group([(fq, g(e + 'attn.wq.weight'), None), (fk, g(e + 'attn.wk.weight'), None), (fv, g(e + 'attn.wv.weight'), None)])
        if FUSED_SELF_ATTENTION:        # vertex queries on joint keys through the same fused core (Tk = J)
            o = ops.attention(q, k, v, Hh, d ** -0.5, rates.mdr_attn, gen, training)
        else:
            qh, kh, vh = [ops.reshape(t, B, t.shape[1], Hh, d).permute(0, 2, 1, 3) for t in (q, k, v)]
            att = ops.dropout(ops.softmax(ops.matmul(qh, kh.transpose(-2, -1), d ** -0.5)), rates.mdr_attn, gen, training)
            o = ops.contiguous(ops.matmul(att, vh).transpose(1, 2)).reshape(B, V, E)
        o = ops.linear(o, g(e + 'attn.proj.weight'), g(e + 'attn.proj.bias'))
        if FUSED_SELF_ATTENTION:
            vf = ops.drop_fused(o, res, False, rates.mdr_drop, rates.mdr_path, gen, training)           # MDR.py:66
        else:
            vf = ops.add(res, ops.drop_path(ops.dropout(o, rates.mdr_drop, gen, training), rates.mdr_path, gen, training))
        y, res = ops.layernorm_skip(vf, g(e + 'norm2.weight'), g(e + 'norm2.bias'), 1e-5, 0)
        if FUSED_SELF_ATTENTION:                                                                         # timm Mlp
            h = ops.drop_fused(ops.linear(y, g(e + 'mlp.fc1.weight'), g(e + 'mlp.fc1.bias')), None, True, rates.mdr_drop, 0.0, gen, training)
            vf = ops.drop_fused(ops.linear(h, g(e + 'mlp.fc2.weight'), g(e + 'mlp.fc2.bias')), res, False, rates.mdr_drop, rates.mdr_path, gen, training)
        else:
            h = ops.dropout(ops.gelu(ops.linear(y, g(e + 'mlp.fc1.weight'), g(e + 'mlp.fc1.bias'))), rates.mdr_drop, gen, training)
            h = ops.dropout(ops.linear(h, g(e + 'mlp.fc2.weight'), g(e + 'mlp.fc2.bias')), rates.mdr_drop, gen, training)
            vf = ops.add(res, ops.drop_path(h, rates.mdr_path, gen, training))
        vf = ops.layernorm(vf, g('norm%s.a_2' % sfx), g('norm%s.b_2' % sfx), 1e-6, 1)          # vanilla_transformer_encoder.py:31-34
        sa = 'selfatt%s.linears.' % sfx
        vf, res, xq, xk = ops.fork(vf, 4)
        qq, kk, vv = ops.linear_group([(t, g(sa + '%d.weight' % n), g(sa + '%d.bias' % n)) for n, t in enumerate((vf, xq, xk))])   # [B,431,64] each, one launch
        if FUSED_SELF_ATTENTION:        # one launch forward / three backward, no [B,2,431,431] tensor (csrc/train_attn.inc)
            xo = ops.attention(qq, kk, vv, Hh, 1.0 / math.sqrt(d), rates.mdr_self, gen, training)
        else:                           # the same core composed from the primitives (the cross-check of tests/test_gpu_train_fused.py)
            qh, kh, vh = [ops.reshape(t, B, V, Hh, d).transpose(1, 2) for t in (qq, kk, vv)]
            pa = ops.dropout(ops.softmax(ops.matmul(qh, kh.transpose(-2, -1), 1.0 / math.sqrt(d))), rates.mdr_self, gen, training)
            xo = ops.contiguous(ops.matmul(pa, vh).transpose(1, 2)).reshape(B, V, E)
        xo = ops.linear(xo, g(sa + '3.weight'), g(sa + '3.bias'))
        vf = ops.drop_fused(xo, res, False, rates.mdr_self, 0.0, gen, training) if FUSED_SELF_ATTENTION else \
            ops.add(res, ops.dropout(xo, rates.mdr_self, gen, training))                         # MDR.py:143
    # MDR head (MDR.py:156-168)
    va, vb, vs = ops.fork(vf, 3)
    ac = ops.linear(va, g('motion_linear.weight'), g('motion_linear.bias'))
    mat_a, mat_c = ops.split(ac, 2, (20, 3))
    mat_b = ops.linear(vb, g('bias_linear.weight'), g('bias_linear.bias'))
    if c.alpha:
        mat_b = ops.layernorm(mat_b, g('bias_norm.weight'), g('bias_norm.bias'), 1e-5, 0)
    elif training:
        rm = buffers.get(p + 'bias_norm.running_mean') if buffers is not None else None
        rv = buffers.get(p + 'bias_norm.running_var') if buffers is not None else None
        if FUSED_SELF_ATTENTION:        # one launch per direction, one workgroup per vertex channel
            mat_b = ops.batchnorm_train(mat_b, g('bias_norm.weight'), g('bias_norm.bias'), rm, rv)
        else:
            mat_b = _batchnorm_train(mat_b, g('bias_norm.weight'), g('bias_norm.bias'), rm, rv)
    else:
        rm, rv = buffers[p + 'bias_norm.running_mean'].reshape(1, -1, 1), buffers[p + 'bias_norm.running_var'].reshape(1, -1, 1)
        xh = ops.mul(ops.sub(mat_b, rm), ops.raw_unary(ops.U_RSQRT, ops.raw_unary(ops.U_AFFINE, rv, 1.0, 1e-5)))
        mat_b = ops.add(ops.mul(xh, g('bias_norm.weight').reshape(1, -1, 1)), g('bias_norm.bias').reshape(1, -1, 1))
    mat_b = conv1d_k3(ops.gelu(mat_b), g('bias_conv1d.weight'), g('bias_conv1d.bias'))         # [B,20,3]
    mix = ops.matmul(ops.softmax(mat_a), mat_b)                                                 # [B,431,3]
    if c.alpha:
        mix = ops.mul(ops.pow_base(1.1, ops.linear(vs, g('scale_linear.weight'), g('scale_linear.bias'))), mix)
    vc = ops.add(mix, mat_c)
    out = conv1d_k3(vc, g('upsample_conv.weight'), g('upsample_conv.bias'))                    # [B,6890,3] (a transposed view)
    return ops.add(out, c.v6890)


def gator_forward(P, c, pose2d, gen=None, rates=None, training=True, buffers=None):
    """GATOR.forward (lib/models/GATOR.py:16-22): pose2d [B,J,2] -> (cam_mesh [B,6890,3] m, pose3d [B,J,3] mm), differentiable."""
    gen = gen or ops.Generator(0)
    rates = rates or Rates(0.0)
    B, J = pose2d.shape[0], c.J
    x_out, feat = gat_forward(P, c, pose2d, gen, rates, training)
    pose3d, p3 = ops.fork(x_out.reshape(B, J, 3))
    pc = ops.cat([pose2d, ops.affine(p3, 1.0 / 1000.0), feat], 2)
    mesh = mdr_forward(P, c, pc, gen, rates, training, buffers)
    return mesh, pose3d
