"""The primitives of the training row (include/gator_train.h, gator_amd/train/ops.py) against torch-CPU float64 autograd of the
same operation: value and every input gradient, on strided / broadcast / batched operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gator_amd.train import ops

pytestmark = pytest.mark.gpu


def _check(ours, ref, inputs, tol=2e-5, seed=0):
    """inputs: list of float64 CPU tensors (requires_grad as wanted).  Compares value and gradients, relative to max|ref|."""
    rs = np.random.RandomState(seed)
    dev = [t.detach().float().cuda().requires_grad_(t.requires_grad) for t in inputs]
    y = ours(*dev)
    yr = ref(*inputs)
    assert tuple(y.shape) == tuple(yr.shape), (y.shape, yr.shape)
    w = torch.from_numpy(rs.randn(*yr.shape)) if yr.dim() else torch.tensor(1.0, dtype=torch.float64)
    scale = max(1e-30, float(yr.detach().abs().max()))
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) <= tol * scale, 'value'
    need = [t for t in inputs if t.requires_grad]
    gr = torch.autograd.grad((yr * w).sum(), need)
    go = torch.autograd.grad(y, [d for d in dev if d.requires_grad], grad_outputs=w.float().cuda())
    for i, (a, b) in enumerate(zip(go, gr)):
        assert tuple(a.shape) == tuple(b.shape)
        s = max(1e-30, float(b.abs().max()))
        err = float((a.cpu().double() - b).abs().max())
        assert err <= tol * s, 'grad %d: %.3e vs scale %.3e' % (i, err, s)


def _r(*shape, seed=0, grad=True):
    return torch.from_numpy(np.random.RandomState(seed).randn(*shape)).requires_grad_(grad)


@pytest.mark.parametrize('name', ['add', 'sub', 'mul', 'div'])
def test_binary_broadcast(name):
    a, b = _r(3, 5, 7, seed=1), _r(5, 1, seed=2)
    if name == 'div':
        b = (b.detach().abs() + 0.5).requires_grad_(True)
    ref = {'add': torch.add, 'sub': torch.sub, 'mul': torch.mul, 'div': torch.div}[name]
    _check(getattr(ops, name), ref, [a, b])
    # strided operand (a transposed view) and a scalar-like broadcast
    _check(lambda x, y: getattr(ops, name)(x.transpose(0, 2), y), lambda x, y: ref(x.transpose(0, 2), y), [_r(7, 5, 3, seed=3), b])


@pytest.mark.parametrize('name,ref', [('gelu', F.gelu), ('exp', torch.exp), ('abs_', torch.abs), ('square', lambda x: x * x)])
def test_unary(name, ref):
    _check(getattr(ops, name), ref, [_r(4, 33, 5, seed=4)])


def test_unary_positive_domain():
    x = (_r(6, 40, seed=5).detach().abs() + 0.3).requires_grad_(True)
    _check(ops.rsqrt, torch.rsqrt, [x])
    _check(ops.sqrt, torch.sqrt, [x])
    _check(ops.recip, torch.reciprocal, [x])
    _check(lambda t: ops.pow_base(1.1, t), lambda t: 1.1 ** t, [_r(6, 40, seed=6)])
    _check(lambda t: ops.affine(t, -2.5, 0.75), lambda t: -2.5 * t + 0.75, [_r(6, 40, seed=7)])


@pytest.mark.parametrize('shape,dims', [((7, 300, 5), (0,)), ((7, 300, 5), (1,)), ((7, 300, 5), (0, 2)), ((5000, 12), (0,)), ((3, 70000), (1,)),
                                        ((2, 3, 4, 5), (0, 1, 2, 3)), ((64, 431, 3), (0, 2))])
def test_sum(shape, dims):
    _check(lambda t: ops.sum_(t, dims), lambda t: t.sum(dims), [_r(*shape, seed=8)], tol=1e-5)
    _check(lambda t: ops.mean(t, dims, keepdim=True), lambda t: t.mean(dims, keepdim=True), [_r(*shape, seed=9)], tol=1e-5)


@pytest.mark.parametrize('sa,sb', [((70, 33), (33, 90)), ((5, 17, 16), (5, 16, 17)), ((3, 8, 17, 16), (3, 8, 16, 17)), ((17, 17), (6, 17, 128)),
                                   ((4, 431, 64), (64, 20)), ((2, 2, 431, 32), (2, 2, 32, 431))])
def test_matmul(sa, sb):
    _check(lambda a, b: ops.matmul(a, b, 0.25), lambda a, b: 0.25 * (a @ b), [_r(*sa, seed=10), _r(*sb, seed=11)])


def test_matmul_strided_heads():
    """the GAT attention access pattern: q, k, v are strided slices of one qkv tensor [B,J,3,H,16]"""
    qkv = _r(4, 17, 3 * 8 * 16, seed=12)

    def f(t, mm, sm):
        z = t.reshape(4, 17, 3, 8, 16).permute(2, 0, 3, 1, 4)
        att = sm(mm(z[0], z[1].transpose(-2, -1)))
        return mm(att, z[2])

    _check(lambda t: f(t, lambda a, b: ops.matmul(a, b), ops.softmax), lambda t: f(t, torch.matmul, lambda s: s.softmax(-1)), [qkv])


def test_linear_with_split_k():
    x, w, b = _r(6000, 64, seed=13), _r(48, 64, seed=14), _r(48, seed=15)
    _check(ops.linear, F.linear, [x, w, b])
    _check(lambda a, c: ops.linear(a, c), lambda a, c: F.linear(a, c), [_r(3, 19, 128, seed=16), _r(51, 128, seed=17)])


@pytest.mark.parametrize('n', [3, 17, 64, 128, 431])
def test_softmax(n):
    _check(ops.softmax, lambda t: t.softmax(-1), [_r(5, 9, n, seed=18)])


@pytest.mark.parametrize('n', [3, 64, 128, 304])
def test_layernorm(n):
    x, w, b = _r(37, n, seed=19), _r(n, seed=20), _r(n, seed=21)
    _check(lambda a, c, d: ops.layernorm(a, c, d, 1e-5, 0), lambda a, c, d: F.layer_norm(a, (n,), c, d, 1e-5), [x, w, b])
    _check(lambda a: ops.layernorm(a, None, None, 1e-5, 0), lambda a: F.layer_norm(a, (n,), None, None, 1e-5), [x])

    def custom(a, c, d):                                   # lib/models/vanilla_transformer_encoder.py:31-34
        return c * (a - a.mean(-1, keepdim=True)) / (a.std(-1, keepdim=True) + 1e-6) + d
    _check(lambda a, c, d: ops.layernorm(a, c, d, 1e-6, 1), custom, [x, w, b])


def test_cat_narrow_fork_contiguous():
    a, b = _r(3, 5, 4, seed=22), _r(3, 2, 4, seed=23)
    _check(lambda x, y: ops.cat([x, y], 1), lambda x, y: torch.cat([x, y], 1), [a, b])
    _check(lambda x: ops.narrow(x, 1, 1, 3), lambda x: x.narrow(1, 1, 3), [a])
    _check(lambda x: ops.contiguous(x.permute(2, 0, 1)), lambda x: x.permute(2, 0, 1).contiguous(), [a])

    def f(x, mul_, add_, fork_):
        p, q = fork_(x)
        return add_(mul_(p, p), q)
    _check(lambda x: f(x, ops.mul, ops.add, ops.fork), lambda x: f(x, torch.mul, torch.add, lambda t: (t, t)), [a])


def test_dropout_statistics_and_backward():
    g = ops.Generator(123)
    x = torch.ones(1 << 18, device='cuda', requires_grad=True)
    y = ops.dropout(x, 0.4, g)
    keep = float((y.detach() != 0).float().mean())
    assert abs(keep - 0.6) < 0.01
    assert torch.all((y.detach() == 0) | ((y.detach() - 1 / 0.6).abs() < 1e-6))
    gx, = torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y))
    assert torch.equal(gx, y.detach())                      # same mask, same scale
    y2 = ops.dropout(x, 0.4, ops.Generator(123))
    assert torch.equal(y2.detach(), y.detach())             # reproducible from (seed, offset)
    y3 = ops.dropout(x, 0.4, g)
    assert not torch.equal(y3.detach(), y.detach())         # a new offset draws new numbers
    assert ops.dropout(x, 0.4, g, training=False) is x
    z = ops.drop_path(torch.ones(4096, 3, 5, device='cuda'), 0.2, g)
    per = z.reshape(4096, -1)
    assert torch.all((per == per[:, :1]).all(1))            # one decision per sample
    assert abs(float((per[:, 0] != 0).float().mean()) - 0.8) < 0.03


def test_linear_group_matches_separate_linears():
    xs = [_r(5, 431, 64, seed=30), _r(5, 17, 64, seed=31), _r(5, 17, 64, seed=32)]
    ws = [_r(64, 64, seed=33), _r(64, 64, seed=34), _r(48, 64, seed=35)]
    bs = [_r(64, seed=36), None, _r(48, seed=37)]
    ins = xs + ws + [b for b in bs if b is not None]

    def ours(x0, x1, x2, w0, w1, w2, b0, b2):
        o = ops.linear_group([(x0, w0, b0), (x1, w1, None), (x2, w2, b2)])
        return ops.cat([o[0].reshape(5, -1), o[1].reshape(5, -1), o[2].reshape(5, -1)], 1)

    def ref(x0, x1, x2, w0, w1, w2, b0, b2):
        return torch.cat([F.linear(x0, w0, b0).reshape(5, -1), F.linear(x1, w1).reshape(5, -1), F.linear(x2, w2, b2).reshape(5, -1)], 1)

    _check(ours, ref, ins)


def test_gradient_slot_may_feed_only_one_slot_aware_op():
    """the in-place weight-gradient slots were safe only by convention.  A parameter view used by two slot-aware ops
    in one step (both would write the same slice of the flat gradient, deferred, accumulate = 0) now raises instead of silently
    losing a gradient; disjoint parts of one view (the two rows of the MGCN weight) stay legal."""
    w = torch.randn(8, 16, device='cuda', requires_grad=True)
    w._gslot = torch.zeros(8, 16, device='cuda')
    x = torch.randn(4, 16, device='cuda')
    ops.linear(x, w, None)
    with pytest.raises(RuntimeError, match='two gradient-slot-aware ops'):
        ops.linear(x, w, None)
    W = torch.randn(2, 16, 16, device='cuda', requires_grad=True)
    W._gslot = torch.zeros(2, 16, 16, device='cuda')
    ops.xw(x, W, 0)
    ops.xw(x, W, 1)
    with pytest.raises(RuntimeError, match='two gradient-slot-aware ops'):
        ops.xw(x, W, 1)
