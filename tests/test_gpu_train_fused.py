"""The fused training ops (attention core, small J x J attention, MGCN aggregation, dropout / DropPath chains, BatchNorm) against the same
thing composed from the primitives: forward values and every gradient, with dropout ON (both draw the same Philox masks from the same
generator state)."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from gator_amd.train import model as M, ops
from tests.test_gpu_train_step import make_trainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,T,Tk,rate', [(3, 431, 431, 0.0), (5, 431, 431, 0.1), (2, 77, 77, 0.3), (4, 431, 17, 0.2), (3, 431, 19, 0.0)])
def test_fused_attention_matches_composed(B, T, Tk, rate):
    """ops.attention (one launch forward, three backward, no [B,H,T,T] tensor) against softmax -> dropout -> matmul composed from the
    primitives with the same generator state: output and dq, dk, dv."""
    rs = np.random.RandomState(B * 1000 + T)
    H, D = 2, 32
    q, k, v = [torch.from_numpy(rs.randn(B, n, H * D).astype(np.float32)).cuda().requires_grad_(True) for n in (T, Tk, Tk)]
    w = torch.from_numpy(rs.randn(B, T, H * D).astype(np.float32)).cuda()
    scale = 1.0 / np.sqrt(D)

    def composed(q, k, v, gen):
        qq, kk, vv = [ops.reshape(t, B, t.shape[1], H, D).transpose(1, 2) for t in (q, k, v)]
        pa = ops.dropout(ops.softmax(ops.matmul(qq, kk.transpose(-2, -1), scale)), rate, gen, True)
        return ops.contiguous(ops.matmul(pa, vv).transpose(1, 2)).reshape(B, T, H * D)

    want = composed(q, k, v, ops.Generator(3))
    gw = torch.autograd.grad(want, [q, k, v], grad_outputs=w)
    got = ops.attention(q, k, v, H, scale, rate, ops.Generator(3), True)
    gg = torch.autograd.grad(got, [q, k, v], grad_outputs=w)
    for nm, a, b in [('o', got, want)] + [('d' + n, x, y) for n, x, y in zip('qkv', gg, gw)]:
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        print('%s: max|fused - composed| %.2e (scale %.2e)' % (nm, err, sc))
        assert err <= 2e-5 * sc, nm


@pytest.mark.parametrize('J,rate', [(17, 0.0), (19, 0.4)])
def test_small_attention_matches_composed(J, rate):
    """ops.attention_small (GAT encoder attention, one wave per (sample, head)) against matmul -> +bias -> softmax -> dropout -> matmul."""
    rs = np.random.RandomState(J)
    B, H, D = 7, 8, 16
    C = H * D
    qkv = torch.from_numpy(rs.randn(B, J, 3 * C).astype(np.float32)).cuda().requires_grad_(True)
    bias = torch.from_numpy(rs.randn(H, J, J).astype(np.float32)).cuda().requires_grad_(True)
    w = torch.from_numpy(rs.randn(B, J, C).astype(np.float32)).cuda()

    def composed(qkv, bias, gen):
        q, k, v = [ops.reshape(t, B, J, H, D).permute(0, 2, 1, 3) for t in ops.split(qkv.reshape(B, J, 3, C), 2, (1, 1, 1))]
        att = ops.dropout(ops.softmax(ops.add(ops.matmul(q, k.transpose(-2, -1), 0.25), bias)), rate, gen, True)
        return ops.contiguous(ops.matmul(att, v).transpose(1, 2)).reshape(B, J, C)

    want = composed(qkv, bias, ops.Generator(9))
    gw = torch.autograd.grad(want, [qkv, bias], grad_outputs=w)
    got = ops.attention_small(qkv, bias, H, 0.25, rate, ops.Generator(9), True)
    gg = torch.autograd.grad(got, [qkv, bias], grad_outputs=w)
    for nm, a, b in (('o', got, want), ('dqkv', gg[0], gw[0]), ('dbias', gg[1], gw[1])):
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        print('%s: max|fused - composed| %.2e (scale %.2e)' % (nm, err, sc))
        assert err <= 2e-5 * sc, nm


def test_fused_mgcn_matches_composed():
    rs = np.random.RandomState(4)
    B, J, C = 6, 19, 128
    t = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32)).cuda().requires_grad_(True)
    h0, h1, adj, Mm, bias = t(B, J, C), t(B, J, C), t(J, J), t(J, C), t(C)
    w = torch.from_numpy(rs.randn(B, J, C).astype(np.float32)).cuda()
    E = torch.eye(J, device='cuda')

    def composed(h0, h1, adj, Mm, bias):
        a1, a2 = ops.fork(adj)
        m1, m2 = ops.fork(Mm)
        return ops.add(ops.add(ops.matmul(ops.mul(a1, E), ops.mul(m1, h0)), ops.matmul(ops.mul(a2, 1.0 - E), ops.mul(m2, h1))), bias.reshape(1, 1, -1))

    want = composed(h0, h1, adj, Mm, bias)
    gw = torch.autograd.grad(want, [h0, h1, adj, Mm, bias], grad_outputs=w)
    got = ops.mgcn(h0, h1, adj, Mm, bias)
    gg = torch.autograd.grad(got, [h0, h1, adj, Mm, bias], grad_outputs=w)
    for nm, a, b in [('out', got, want)] + list(zip(('dh0', 'dh1', 'dadj', 'dM', 'dbias'), gg, gw)):
        err, sc = float((a - b).abs().max()), float(b.abs().max())
        print('%s: max|fused - composed| %.2e (scale %.2e)' % (nm, err, sc))
        assert err <= 2e-5 * sc, nm


@pytest.mark.parametrize('gelu,rate,path,with_res', [(True, 0.1, 0.0, False), (False, 0.2, 0.2, True), (False, 0.0, 0.3, True), (True, 0.0, 0.0, False)])
def test_drop_fused_matches_composed(gelu, rate, path, with_res):
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.randn(9, 19, 128).astype(np.float32)).cuda().requires_grad_(True)
    res = torch.from_numpy(rs.randn(9, 19, 128).astype(np.float32)).cuda().requires_grad_(True) if with_res else None
    w = torch.from_numpy(rs.randn(9, 19, 128).astype(np.float32)).cuda()

    def composed(gen):
        v = ops.gelu(x) if gelu else x
        v = ops.drop_path(ops.dropout(v, rate, gen, True), path, gen, True)
        return ops.add(res, v) if with_res else v

    want = composed(ops.Generator(2))
    got = ops.drop_fused(x, res, gelu, rate, path, ops.Generator(2), True)
    ins = [x] + ([res] if with_res else [])
    gw = torch.autograd.grad(want, ins, grad_outputs=w)
    gg = torch.autograd.grad(got, ins, grad_outputs=w)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    for a, b in zip(gg, gw):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())


def test_fused_batchnorm_matches_torch_training_mode():
    import torch.nn.functional as F
    rs = np.random.RandomState(12)
    B, C, L = 9, 431, 3
    x = torch.from_numpy(rs.randn(B, C, L)).requires_grad_(True)
    w, b = torch.from_numpy(rs.randn(C)).requires_grad_(True), torch.from_numpy(rs.randn(C)).requires_grad_(True)
    rm0, rv0 = torch.from_numpy(rs.randn(C)), torch.from_numpy(rs.rand(C) + 0.5)
    gw = torch.from_numpy(rs.randn(B, C, L))
    rm, rv = rm0.clone(), rv0.clone()
    want = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
    gr = torch.autograd.grad(want, [x, w, b], grad_outputs=gw)
    xd, wd, bd = [t.detach().float().cuda().requires_grad_(True) for t in (x, w, b)]
    rmd, rvd = rm0.float().cuda(), rv0.float().cuda()
    got = ops.batchnorm_train(xd, wd, bd, rmd, rvd)
    gg = torch.autograd.grad(got, [xd, wd, bd], grad_outputs=gw.float().cuda())
    assert float((got.detach().cpu().double() - want.detach()).abs().max()) <= 2e-5 * float(want.detach().abs().max())
    for a, r in zip(gg, gr):
        assert float((a.cpu().double() - r).abs().max()) <= 2e-5 * float(r.abs().max())
    assert float((rmd.cpu().double() - rm).abs().max()) <= 1e-6 and float((rvd.cpu().double() - rv).abs().max()) <= 1e-5     # running statistics as torch updates them
