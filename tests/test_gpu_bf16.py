"""BASELINE config 3: the GATOR forward in 16-bit operand mode (gator_forward_bf16).  Round 5: the three MDR layers and the head
features run on ONE fp16 activation plane (mdr_fused.hip, XA = 3: weights on two planes, fp32 accumulate / softmax / norms / GELU /
residual stream); the encoder and the vertex regressor keep the fp32 configuration's two-plane operands (a single 16-bit plane THERE is
what costs millimetres: tools/emulate_16bit.py, profiles/r05_emulate_16bit.txt).  Bar (round-4 review): vertices within 1 mm max /
0.2 mm rms of the fp64 oracle, |delta MPJPE| <= 0.05 mm.  The round-4 form (bf16 vertex regressor only, 8 mm / 1 mm) stays reachable
through GATOR_C3_MDR=0 GATOR_C3_UPSAMPLE_BF16=1 and through the stage entry point tested at the bottom."""
import numpy as np
import pytest
import torch

from gator_amd import eval as geval
from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,B', [('coco19_alpha', 40), ('h36m17_bn', 33), ('coco19_alpha', 1)])
def test_bf16_forward_mpjpe_parity(name, B):
    from oracle import gator_oracle as go
    z, m = build_model(name, 'fused')
    zz, c, sd = oracle_setup(name)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=21))
    ref, rp = go.gator_forward(sd, c, x, torch.float64)
    m.precision = 'bf16'
    verts, pose3d = m(x.cuda())
    m.precision = 'f32'
    v32, _ = m(x.cuda())
    err = np.abs(verts.cpu().numpy().astype(np.float64) - ref.numpy()) * 1e3
    print('\n[%s bf16 B=%d] vertex err vs fp64: max %.3f mm  rms %.3f mm ; fp32 path max %.2e mm'
          % (name, B, err.max(), np.sqrt((err ** 2).mean()), np.abs(v32.cpu().numpy() - ref.numpy()).max() * 1e3))
    assert err.max() < 1.0 and np.sqrt((err ** 2).mean()) < 0.2      # one fp16 activation plane in the MDR layers: 2^-12 relative per operand element
    assert np.abs(pose3d.cpu().numpy() - rp.numpy()).max() <= 1.0           # pose3d in mm: the encoder's linears run on one activation plane too
    jr = synthetic.load_j_regressors()['h36m']
    reg = geval.JointRegressor(jr, 'cuda')
    j_bf = reg(verts * 1000).cpu().numpy()
    j_ref = go.regress_joints(jr, ref * 1000).numpy()
    assert np.abs(j_bf - j_ref).max() < 1.0                                  # joints average ~6 vertices each
    gt = j_ref + np.random.RandomState(0).randn(*j_ref.shape) * 30.0
    e_bf = go.mpjpe(j_bf, gt, list(geval.H36M_EVAL_JOINTS))
    e_ref = go.mpjpe(j_ref, gt, list(geval.H36M_EVAL_JOINTS))
    print('[%s bf16] joints max %.3f mm, MPJPE %.4f vs %.4f mm' % (name, np.abs(j_bf - j_ref).max(), e_bf, e_ref))
    assert abs(e_bf - e_ref) < 0.05


def test_bf16_upsample_stage():
    from oracle import gator_oracle as go
    z, m = build_model('h36m17_bn', 'fused')
    zz, c, sd = oracle_setup('h36m17_bn')
    taps = {}
    ref, _ = go.gator_forward(sd, c, torch.from_numpy(z['pose2d']), torch.float64, taps)
    mdr = m.pose2mesh
    mdr.impl = 'fused'
    v = mdr.upsample(taps['vert431'].float().cuda(), precision='bf16')
    err = np.abs(v.cpu().numpy() - ref.numpy()) * 1e3
    assert err.max() < 8.0
    v2 = mdr.upsample(taps['vert431'].float().cuda(), precision='f32')
    assert np.abs(v2.cpu().numpy() - ref.numpy()).max() * 1e3 <= 1e-3


@pytest.mark.timeout(1500)
def test_config3_2048_samples_every_coordinate():
    """BASELINE config 3 at its own size (B = 2048, J = 19), every coordinate of every sample against the fp64 oracle: max <= 1 mm,
    rms <= 0.2 mm, |delta MPJPE| <= 0.05 mm (the round-4 review's bar for "config 3 is real"); deterministic; pose3d is the fp32 path's."""
    from oracle import gator_oracle as go
    B, J = 2048, 19
    z, m = build_model('coco19_alpha', 'fused')
    zz, c, sd = oracle_setup('coco19_alpha')
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=77))
    m.precision = 'bf16'
    v, p = m(x.cuda())
    v2, _ = m(x.cuda())
    assert torch.equal(v, v2)
    m.precision = 'f32'
    vf, pf = m(x.cuda())
    assert float((p - pf).abs().max()) <= 1.0                       # pose3d (mm) of the 16-bit encoder against the fp32 path's
    m.device_status()
    jr = synthetic.load_j_regressors()['h36m']
    worst, sq, n, jd = 0.0, 0.0, 0, []
    gt_rs = np.random.RandomState(0)
    e16, e64 = [], []
    for s in range(0, B, 256):
        ref, _ = go.gator_forward(sd, c, x[s:s + 256], torch.float64)
        e = np.abs(v[s:s + 256].cpu().numpy().astype(np.float64) - ref.numpy()) * 1e3
        worst = max(worst, float(e.max()))
        sq += float((e ** 2).sum())
        n += e.size
        j16 = go.regress_joints(jr, v[s:s + 256].cpu().double() * 1000).numpy()
        j64 = go.regress_joints(jr, ref * 1000).numpy()
        gt = j64 + gt_rs.randn(*j64.shape) * 30.0
        e16.append(go.mpjpe(j16, gt, list(geval.H36M_EVAL_JOINTS)))
        e64.append(go.mpjpe(j64, gt, list(geval.H36M_EVAL_JOINTS)))
    rms = float(np.sqrt(sq / n))
    dm = abs(float(np.mean(e16)) - float(np.mean(e64)))
    d32 = float((v - vf).abs().max()) * 1e3
    print('\n[config 3, B=2048 J=19, %d coordinates] vs fp64 oracle: max %.3f mm  rms %.4f mm  |dMPJPE| %.5f mm ; vs the fp32 path max %.3f mm'
          % (n, worst, rms, dm, d32))
    assert worst <= 1.0 and rms <= 0.2 and dm <= 0.05


@pytest.mark.parametrize('gain', [3.0, 8.0])
def test_config3_attention_long_way_under_large_logits(gain, monkeypatch):
    """The one-plane attention proves the fp16 range of its probabilities by their row sums and redoes a tile the long way (maximum,
    rescale, new reference) where that fails; with the shipped synthetic weights only the first tile of a head does.  Here the q / k
    projections of all three layers are scaled so that the logits grow 9 x / 64 x: tiles jump by more than 2^9 all the time.  The
    result must stay finite, deterministic, batch independent and inside the mode's bar against the fp64 oracle of the same weights."""
    from oracle import gator_oracle as go
    z, m = build_model('h36m17_bn', 'fused', device=None)
    zz, c, sd_o = oracle_setup('h36m17_bn')
    sd = m.state_dict()
    for sfx in ('', '_1', '_2'):
        for n in (0, 1):
            for leaf in ('weight', 'bias'):
                k = 'pose2mesh.selfatt%s.linears.%d.%s' % (sfx, n, leaf)
                sd[k] = sd[k] * gain
                sd_o[k] = sd_o[k] * gain
    m.load_state_dict(sd)
    monkeypatch.setenv('GATOR_C3_GUARD', '0')      # this test is about the long-way path of the one-plane loop: the guard (next test) would take gain 8 out of the mode
    m = m.cuda()
    m.precision = 'bf16'
    x = torch.from_numpy(synthetic.synthetic_pose2d(24, 17, seed=9))
    v, p = m(x.cuda())
    torch.cuda.synchronize()
    m.device_status()
    assert m.c3_state()[0]
    assert torch.isfinite(v).all()
    r64, _ = go.gator_forward(sd_o, c, x, torch.float64)
    e = np.abs(v.cpu().numpy().astype(np.float64) - r64.numpy()) * 1e3
    print('\n[config 3, logit gain %.0f^2] vs fp64: max %.3f mm rms %.4f mm' % (gain, e.max(), np.sqrt((e ** 2).mean())))
    # gain 3 (logits x 9): inside the mode's bar.  gain 8 (logits x 64, near one-hot softmaxes): one fp16 plane moves a score by 2^-12 of
    # its magnitude, i.e. the probabilities by percents -- the 16-bit mode is not for such weights (measured 7.6 mm max / 0.64 mm rms; the
    # fp32 mode stays inside twice the reference's own noise: test_gpu_hardening.py); what must still hold is finite, deterministic,
    # batch-independent output from the long-way path.
    if gain <= 3.0:
        assert e.max() < 1.0 and np.sqrt((e ** 2).mean()) < 0.2
    else:
        # Against the fp32 mode on the SAME weights (round-5 review: a stated ratio instead of "< 20 mm"): one plane carries 11 significant
        # bits of a score where the fp32 mode's two planes carry 22, so the 16-bit mode may sit up to 2^11 above it -- and no further.
        m.precision = 'f32'
        v32, _ = m(x.cuda())
        m.precision = 'bf16'
        e32 = np.abs(v32.cpu().numpy().astype(np.float64) - r64.numpy()) * 1e3
        r_max, r_rms = e.max() / e32.max(), np.sqrt((e ** 2).mean()) / np.sqrt((e32 ** 2).mean())
        print('[config 3, logit gain %.0f^2] fp32 mode on the same weights: max %.2e mm rms %.2e mm -> 16-bit / fp32 = %.0f (max), %.0f (rms); bound 2^11 = 2048'
              % (gain, e32.max(), np.sqrt((e32 ** 2).mean()), r_max, r_rms))
        assert r_max <= 2048.0 and r_rms <= 2048.0
    v2, _ = m(x.cuda())
    vs, _ = m(x[5:9].cuda())
    assert torch.equal(v, v2) and torch.equal(vs, v[5:9])


@pytest.mark.parametrize('gain,one_plane', [(1.0, True), (3.0, True), (8.0, False)])
def test_config3_guard_keeps_two_planes_under_sharp_attention(gain, one_plane):
    """Round 6: gator_create bounds the self-attention logits from the weights (sigma_max(Wq_h^T Wk_h) x the custom LayerNorm's bound on
    |x|^2); above 2^10 in the exp2 domain the MDR layers of gator_forward_bf16 keep two planes for that ctx.  The shipped synthetic weights (bound ~50)
    and the logits-x-9 weights (~470, measured 0.63 mm) stay in the mode; logits x 64 (~3 300, measured 7.6 mm in the mode) fall out of it and meet
    the mode's bar again -- a checkpoint with sharp attention degrades in speed, not silently in millimetres."""
    from oracle import gator_oracle as go
    z, m = build_model('h36m17_bn', 'fused', device=None)
    zz, c, sd_o = oracle_setup('h36m17_bn')
    sd = m.state_dict()
    for sfx in ('', '_1', '_2'):
        for n in (0, 1):
            for leaf in ('weight', 'bias'):
                k = 'pose2mesh.selfatt%s.linears.%d.%s' % (sfx, n, leaf)
                sd[k] = sd[k] * gain
                sd_o[k] = sd_o[k] * gain
    m.load_state_dict(sd)
    m = m.cuda()
    m.precision = 'bf16'
    x = torch.from_numpy(synthetic.synthetic_pose2d(24, 17, seed=9))
    v, p = m(x.cuda())
    torch.cuda.synchronize()
    m.device_status()
    state, bound = m.c3_state()
    r64, _ = go.gator_forward(sd_o, c, x, torch.float64)
    e = np.abs(v.cpu().numpy().astype(np.float64) - r64.numpy()) * 1e3
    print('\n[config 3 guard, logit gain %.0f^2] bound %.0f -> MDR layers on %s; vs fp64: max %.3f mm rms %.4f mm' % (gain, bound, 'one plane' if state else 'two planes', e.max(), np.sqrt((e ** 2).mean())))
    assert state == one_plane
    assert (bound <= 1024.0) == one_plane
    assert e.max() < 1.0 and np.sqrt((e ** 2).mean()) < 0.2


def test_config3_persistent_launch_equals_four_launches_bitwise(monkeypatch):
    """The 16-bit MDR layers as persistent launches (chunks of <= 384 samples) and as four per-stage launches run the same tile body on the
    same tiles: bitwise the same vertices, whatever the batch (ragged last chunk, a batch below the persistent threshold)."""
    monkeypatch.setenv('GATOR_MDR_PERSIST', '1')
    z, m1 = build_model('coco19_alpha', 'fused')
    monkeypatch.setenv('GATOR_MDR_PERSIST', '0')
    z, m0 = build_model('coco19_alpha', 'fused')
    m1.precision = m0.precision = 'bf16'
    for B in (7, 300, 801):
        x = torch.from_numpy(synthetic.synthetic_pose2d(B, 19, seed=B)).cuda()
        a, b = m1(x), m0(x)
        assert torch.isfinite(a[0]).all()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), B
    m1.device_status()
    m0.device_status()
