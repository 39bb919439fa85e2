"""The sample-tiled GAT encoder (gat_tiled.hip) and the batch split that uses it.

GATOR_GAT_TILED=1 forces it for every batch size (so that the fp64 oracle can check it at sizes it finishes in seconds), =0 keeps
every batch on the one-sample-per-workgroup kernel, unset = the shipped policy (full rounds of the tiled kernel for batches
>= 1024, remainder on the cheaper of the two).  Checks: parity with the oracle (incl. a ragged last workgroup and one sample),
the reference's recorded block activations, bitwise batch / position independence WITHIN the tiled kernel, agreement of the two
kernels to fp32 noise, and the shipped policy at B=2048 (a tiled prefix + a k_gat remainder in one call)."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup

pytestmark = pytest.mark.gpu


def _model(monkeypatch, name, mode):
    if mode is None:
        monkeypatch.delenv('GATOR_GAT_TILED', raising=False)
    else:
        monkeypatch.setenv('GATOR_GAT_TILED', mode)
    return build_model(name, 'fused')          # a fresh module -> a fresh context, which reads the switch


@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
@pytest.mark.parametrize('B', [1, 6, 7, 8, 45])
def test_tiled_encoder_vs_oracle(monkeypatch, name, B):
    from oracle import gator_oracle as go
    z, m = _model(monkeypatch, name, '1')
    zz, c, sd = oracle_setup(name)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=300 + B))
    ref, rp = go.gator_forward(sd, c, x, torch.float64)
    v, p = m(x.cuda())
    e = float(np.abs(v.cpu().numpy().astype(np.float64) - ref.numpy()).max() * 1e3)
    ep = float(np.abs(p.cpu().numpy().astype(np.float64) - rp.numpy()).max())
    print('\n[%s tiled B=%d] verts %.2e mm, pose3d %.2e mm' % (name, B, e, ep))
    assert e <= 1e-3 and ep <= 1e-3


@pytest.mark.parametrize('h4', ['1', '0'])
@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
def test_tiled_block_taps_and_feat_match_reference(monkeypatch, name, h4):
    monkeypatch.setenv('GATOR_GAT_TILED_H4', h4)      # four partial products (default) | the exact six
    z, m = _model(monkeypatch, name, '1')
    B, J = z['pose2d'].shape[:2]
    x = torch.from_numpy(z['pose2d']).cuda()
    m(x)
    m.enable_block_taps(True)
    v, p = m(x)
    for tap in ('gat_block0', 'gat_block5', 'feat'):
        t = m.get_tap(tap, (B, J, 128)).cpu().numpy().astype(np.float64)
        ref = z[tap].astype(np.float64)
        err = np.abs(t - ref).max()
        print('[%s tiled] %s max|d| %.2e (scale %.2f)' % (name, tap, err, np.abs(ref).max()))
        assert err <= 4e-6 * max(1.0, np.abs(ref).max()), tap
    assert np.abs(v.cpu().numpy().astype(np.float64) - z['verts_f64']).max() * 1e3 <= 1e-3


def test_tiled_is_bitwise_batch_and_position_independent(monkeypatch):
    """Inside the tiled kernel a sample's result does not depend on the batch size, on its slot in a workgroup, or on its
    neighbours: permutations, slices and a 2-way shard of a 100-sample batch are bit-identical."""
    z, m = _model(monkeypatch, 'h36m17_bn', '1')
    x = torch.from_numpy(synthetic.synthetic_pose2d(100, 17, seed=5)).cuda()
    v, p = m(x)
    v2, p2 = m(x)
    assert torch.equal(v, v2) and torch.equal(p, p2)
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(1)).cuda()
    vp, pp = m(x[perm])
    assert torch.equal(vp, v[perm]) and torch.equal(pp, p[perm])
    for lo, hi in ((0, 1), (3, 4), (10, 31), (50, 100)):
        vs, ps = m(x[lo:hi].contiguous())
        assert torch.equal(vs, v[lo:hi]) and torch.equal(ps, p[lo:hi]), (lo, hi)


def test_two_encoders_agree_to_fp32_noise(monkeypatch):
    x = torch.from_numpy(synthetic.synthetic_pose2d(64, 19, seed=9)).cuda()
    z, mt = _model(monkeypatch, 'coco19_alpha', '1')
    vt, pt = mt(x)
    z, m1 = _model(monkeypatch, 'coco19_alpha', '0')
    v1, p1 = m1(x)
    d = float((vt - v1).abs().max()) * 1e3
    print('\ntiled vs one-sample-per-workgroup encoder: max %.2e mm' % d)
    assert d <= 1.5e-3 and float((pt - p1).abs().max()) <= 1e-3


def test_shipped_policy_b2048(monkeypatch):
    """Default policy at B=2048, J=17: 1792 samples on the tiled kernel + 256 on k_gat in one call.  Deterministic; each part
    is bit-identical to the same samples run alone on the same kernel; every sample within 1e-3 mm of the fp64 oracle (sampled)."""
    from oracle import gator_oracle as go
    z, m = _model(monkeypatch, 'h36m17_bn', None)
    zz, c, sd = oracle_setup('h36m17_bn')
    x = torch.from_numpy(synthetic.synthetic_pose2d(2048, 17, seed=123)).cuda()
    v, p = m(x)
    v2, _ = m(x)
    assert torch.equal(v, v2) and torch.isfinite(v).all()
    idx = [0, 5, 1000, 1791, 1792, 1900, 2047]                  # both sides of the split
    ref, rp = go.gator_forward(sd, c, x[idx].cpu(), torch.float64)
    e = np.abs(v[idx].cpu().numpy().astype(np.float64) - ref.numpy()).max() * 1e3
    print('\n[policy B=2048] sampled max %.2e mm' % e)
    assert e <= 1e-3 and np.abs(p[idx].cpu().numpy() - rp.numpy()).max() <= 1e-3
    z, mt = _model(monkeypatch, 'h36m17_bn', '1')
    vt, _ = mt(x[:1792].contiguous())
    assert torch.equal(vt, v[:1792])                             # the tiled prefix
    z, m1 = _model(monkeypatch, 'h36m17_bn', '0')
    v1, _ = m1(x[1792:].contiguous())
    assert torch.equal(v1, v[1792:])                             # the k_gat remainder


def test_config4_single_process_b8192(monkeypatch):
    """BASELINE config 4's single-process counterpart: B=8192 Human3.6M poses in ONE call (what the 8-GPU all-gather of 1024-sample
    shards must reproduce up to the encoder-kernel choice): finite, deterministic, sampled accuracy against the fp64 oracle, and a
    1024-sample shard agrees with the same rows of the big batch to fp32 noise (bitwise when both sit on the tiled encoder)."""
    from oracle import gator_oracle as go
    z, m = _model(monkeypatch, 'h36m17_bn', None)
    zz, c, sd = oracle_setup('h36m17_bn')
    x = torch.from_numpy(synthetic.synthetic_pose2d(8192, 17, seed=2024)).cuda()
    v, p = m(x)
    v2, _ = m(x)
    assert torch.equal(v, v2) and bool(torch.isfinite(v).all())
    idx = [0, 1023, 1024, 4095, 7167, 7168, 8191]
    ref, rp = go.gator_forward(sd, c, x[idx].cpu(), torch.float64)
    assert np.abs(v[idx].cpu().numpy().astype(np.float64) - ref.numpy()).max() * 1e3 <= 1e-3
    vs, ps = m(x[3072:4096].contiguous())                        # rank 3's shard of an 8-way split, unpinned: up to 4 x 256 samples the per-call
    # policy takes the one-sample-per-workgroup encoder, the big batch's rows sit on the tiled one -> fp32 noise, not bits
    assert float((vs - v[3072:4096]).abs().max()) * 1e3 <= 1.5e-3 and float((ps - p[3072:4096]).abs().max()) <= 1e-3
    del v2
    # What ShardedForward does with 8 ranks x 1 024 samples: ONE pinned encoder for every call (gator_set_encoder), so that ALL eight
    # shards - whatever their position - reproduce the rows of the single-process batch bit for bit, also for shard sizes at which
    # the per-call policy would mix the two kernels (8 x 1 000: a single process would run 7 168 tiled + 832 on the per-sample kernel).
    from gator_amd.parallel import ShardedForward
    for n in (1024, 1000):
        for pin in ('tiled', 'sample'):
            m.set_encoder(pin)
            xb = x[:8 * n].contiguous()
            vb, pb = m(xb)
            for r in range(8):
                vs, ps = m(xb[r * n:(r + 1) * n].contiguous())
                assert torch.equal(vs, vb[r * n:(r + 1) * n]) and torch.equal(ps, pb[r * n:(r + 1) * n]), (n, pin, r)
            del vb, pb
    m.set_encoder('auto')
    run = ShardedForward(m, 8, 3, object())                      # (no collective is issued: only the pin is exercised)
    run._pin_encoder(1024)
    assert run._pinned == 'sample'
    run = ShardedForward(m, 8, 3, object())
    run._pin_encoder(1025)
    assert run._pinned == 'tiled'
    vs, ps = m(x[3072:4096].contiguous())
    assert torch.equal(vs, v[3072:4096])
    m.set_encoder('auto')
    del v
