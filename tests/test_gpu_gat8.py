"""The two-role GAT encoder k_gat8 (csrc/gat_roles.hip): four product waves stream the weights and issue the token-wise linears,
four helper waves do everything else.  It is the default one-sample-per-workgroup encoder; GATOR_GAT8=0 keeps k_gat
(csrc/gat_fused.hip) for A/B runs.  Checks: parity with the fp64 oracle (both variants, ragged sizes, one sample), the reference's
recorded block activations, bitwise batch / position independence, agreement with k_gat to fp32 rounding noise, and that the
switch really selects the other kernel (different last bits)."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup

pytestmark = pytest.mark.gpu


def _model(monkeypatch, name, gat8):
    monkeypatch.setenv('GATOR_GAT_TILED', '0')                 # every batch on the one-sample-per-workgroup encoder
    if gat8:
        monkeypatch.delenv('GATOR_GAT8', raising=False)
    else:
        monkeypatch.setenv('GATOR_GAT8', '0')
    return build_model(name, 'fused')          # a fresh module -> a fresh context, which reads the switches


@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
@pytest.mark.parametrize('B', [1, 9, 40])
def test_gat8_vs_oracle(monkeypatch, name, B):
    from oracle import gator_oracle as go
    z, m = _model(monkeypatch, name, True)
    zz, c, sd = oracle_setup(name)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=800 + B))
    ref, rp = go.gator_forward(sd, c, x, torch.float64)
    v, p = m(x.cuda())
    e = float(np.abs(v.cpu().numpy().astype(np.float64) - ref.numpy()).max() * 1e3)
    ep = float(np.abs(p.cpu().numpy().astype(np.float64) - rp.numpy()).max())
    print('\n[%s k_gat8 B=%d] verts %.2e mm, pose3d %.2e mm' % (name, B, e, ep))
    assert e <= 1e-3 and ep <= 1e-3


@pytest.mark.parametrize('h4', ['1', '0'])
@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
def test_gat8_block_taps_and_feat_match_reference(monkeypatch, name, h4):
    monkeypatch.setenv('GATOR_GAT8_H4', h4)           # four partial products + fp16 J x J operators (default) | the exact six
    z, m = _model(monkeypatch, name, True)
    B, J = z['pose2d'].shape[:2]
    x = torch.from_numpy(z['pose2d']).cuda()
    m(x)
    m.enable_block_taps(True)
    v, p = m(x)
    for tap in ('gat_block0', 'gat_block5', 'feat'):
        t = m.get_tap(tap, (B, J, 128)).cpu().numpy().astype(np.float64)
        ref = z[tap].astype(np.float64)
        err = np.abs(t - ref).max()
        print('[%s k_gat8] %s max|d| %.2e (scale %.2f)' % (name, tap, err, np.abs(ref).max()))
        assert err <= 4e-6 * max(1.0, np.abs(ref).max()), tap
    assert np.abs(v.cpu().numpy().astype(np.float64) - z['verts_f64']).max() * 1e3 <= 1e-3


def test_gat8_is_bitwise_batch_and_position_independent(monkeypatch):
    """One workgroup per sample, a fixed order of every sum: permutations, slices and a 3-way shard of a 300-sample batch (more than
    one round of the chip) are bit-identical - the correctness criterion of the multi-GPU all-gather."""
    z, m = _model(monkeypatch, 'h36m17_bn', True)
    x = torch.from_numpy(synthetic.synthetic_pose2d(300, 17, seed=77)).cuda()
    v, p = m(x)
    perm = torch.randperm(300, generator=torch.Generator().manual_seed(1)).cuda()
    v2, p2 = m(x[perm])
    assert torch.equal(v2, v[perm]) and torch.equal(p2, p[perm])
    v3, p3 = m(x[17:118])
    assert torch.equal(v3, v[17:118]) and torch.equal(p3, p[17:118])
    parts = [m(x[a:b]) for a, b in ((0, 100), (100, 200), (200, 300))]
    assert torch.equal(torch.cat([q[0] for q in parts]), v)
    v4, _ = m(x)
    assert torch.equal(v4, v)                                    # run-to-run


@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
def test_gat8_agrees_with_k_gat_to_fp32_noise(monkeypatch, name):
    """Same operand formats and partial products; the MLP's four partial sums group the hidden blocks differently and the value
    bias of the J x J attention is added after P.V instead of before - rounding-level differences only."""
    z, m8 = _model(monkeypatch, name, True)
    J = int(z['num_joint'])
    x = torch.from_numpy(synthetic.synthetic_pose2d(64, J, seed=5)).cuda()
    v8, p8 = m8(x)
    feat8 = m8.get_tap('feat', (64, J, 128)).clone()
    zz, m4 = _model(monkeypatch, name, False)
    v4, p4 = m4(x)
    feat4 = m4.get_tap('feat', (64, J, 128)).clone()
    dv = float((v8 - v4).abs().max()) * 1e3
    dp = float((p8 - p4).abs().max())
    df = float((feat8 - feat4).abs().max())
    print('\n[%s] k_gat8 vs k_gat: verts %.2e mm, pose3d %.2e mm, feat %.2e (scale %.2f)' % (name, dv, dp, df, float(feat4.abs().max())))
    assert dv <= 1.5e-3 and dp <= 1e-3
    assert df <= 2e-5 * max(1.0, float(feat4.abs().max()))
    assert not torch.equal(feat8, feat4)                         # (the switch selected a different kernel)


@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
def test_fused_tail_agrees_with_the_two_tail_launches(monkeypatch, name):
    """Round 6: k_gat8<..., TAIL> runs the lifter (GAT.py:151-152) and the MDR joint tokens / K / V tiles (MDR.py:130-134,37-38,65) as its
    epilogue; GATOR_GAT8_TAIL=0 keeps k_gat_lifter + k_gat_joint (gat_tail.hip), which the sample-tiled encoder still uses.  The two
    agree to fp32 rounding (the lifter sums 128 J exact fp32 products in another order; the joint-token linear's four k blocks are
    partial sums instead of one chain); both sit inside the parity bar; the fused form is bitwise batch / position independent."""
    from oracle import gator_oracle as go
    monkeypatch.setenv('GATOR_GAT_TILED', '0')
    monkeypatch.delenv('GATOR_GAT8_TAIL', raising=False)
    z, m1 = build_model(name, 'fused')
    zz, c, sd = oracle_setup(name)
    B = 70
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=321))
    v1, p1 = m1(x.cuda())
    monkeypatch.setenv('GATOR_GAT8_TAIL', '0')
    z, m0 = build_model(name, 'fused')
    v0, p0 = m0(x.cuda())
    torch.cuda.synchronize()
    m1.device_status(); m0.device_status()
    assert not torch.equal(p1, p0)                          # the switch really selects the other form (different last bits)
    ref, rp = go.gator_forward(sd, c, x[:24], torch.float64)
    for tag, v, p in (('fused tail', v1, p1), ('two launches', v0, p0)):
        e = float(np.abs(v[:24].cpu().numpy().astype(np.float64) - ref.numpy()).max() * 1e3)
        ep = float(np.abs(p[:24].cpu().numpy().astype(np.float64) - rp.numpy()).max())
        print('\n[%s %s] verts %.2e mm, pose3d %.2e mm vs fp64' % (name, tag, e, ep))
        assert e <= 1e-3 and ep <= 1e-3
    assert float((p1 - p0).abs().max()) <= 2e-4 and float((v1 - v0).abs().max()) * 1e3 <= 1e-3
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    vp, pp = m1(x[perm].cuda())
    assert torch.equal(vp, v1[perm.cuda()]) and torch.equal(pp, p1[perm.cuda()])
    vs, ps = m1(x[10:13].cuda())
    assert torch.equal(vs, v1[10:13]) and torch.equal(ps, p1[10:13])
