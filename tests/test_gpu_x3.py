"""Split-precision (X3) stages vs their fp32-input-MFMA counterparts.

The fused path computes its large products on the bf16 MFMA with every fp32 operand split exactly into three bf16 planes
(gator_amd/csrc/x3_common.h).  Each stage keeps the fp32-input MFMA kernel behind an environment switch read when the
context is created (GATOR_GAT_X3 / GATOR_MDR_X3 / GATOR_UPSAMPLE_X3 = 0).  Both forms must meet the same 1e-3 mm bar
against the fp64 oracle, and must agree with each other to fp32 rounding noise."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup

pytestmark = pytest.mark.gpu

SWITCHES = ('GATOR_GAT_X3', 'GATOR_MDR_X3', 'GATOR_UPSAMPLE_X3')      # + GATOR_GAT8_H4: the encoder's linears on 4 products (default) or the exact 6


def _run(monkeypatch, name, x, off, mdr_mode='1', up_mode='1', gat_h4='0'):
    monkeypatch.setenv('GATOR_GAT8_H4', gat_h4)
    monkeypatch.setenv('GATOR_GAT_TILED_H4', gat_h4)
    for k in SWITCHES:
        on = mdr_mode if k == 'GATOR_MDR_X3' else (up_mode if k == 'GATOR_UPSAMPLE_X3' else '1')
        monkeypatch.setenv(k, '0' if k in off else on)
    z, m = build_model(name, 'fused')          # a fresh module -> a fresh context, which reads the switches
    v, p = m(x.cuda())
    torch.cuda.synchronize()
    return v.cpu().numpy().astype(np.float64), p.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize('name,B', [('h36m17_bn', 40), ('coco19_alpha', 33)])
def test_x3_and_fp32_mfma_paths_both_meet_the_bar(monkeypatch, name, B):
    from oracle import gator_oracle as go
    zz, c, sd = oracle_setup(name)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=77))
    ref, rp = go.gator_forward(sd, c, x, torch.float64)
    ref, rp = ref.numpy(), rp.numpy()
    outs = {}
    # 'default' = token-wise linears of the encoder and the MDR layers on four partial products (weights exact, activations on two
    # fp16 planes), both attentions and the vertex regressor on two fp16 planes (GATOR_GAT8_H4=1, GATOR_MDR_X3=2, GATOR_UPSAMPLE_X3=2);
    # 'upsample x3' = the default with the exact three-plane regressor; 'all x3' = no rounded operand anywhere
    for label, off in (('default', ()), ('all x3', ()), ('upsample x3', ()), ('gat fp32', ('GATOR_GAT_X3',)), ('mdr fp32', ('GATOR_MDR_X3',)),
                       ('upsample fp32', ('GATOR_UPSAMPLE_X3',)), ('all fp32', SWITCHES)):
        dflt = label in ('default', 'upsample x3')
        v, p = _run(monkeypatch, name, x, off, '2' if dflt else '1', '2' if label == 'default' else '1', '1' if dflt else '0')
        e = np.abs(v - ref).max() * 1e3
        print('\n[%s B=%d] %-14s max |verts - fp64| = %.2e mm, pose3d %.2e mm' % (name, B, label, e, np.abs(p - rp).max()))
        assert e <= 1e-3, label
        assert np.abs(p - rp).max() <= 1e-3, label
        outs[label] = v
    # the two forms of every stage are the same function up to fp32 rounding noise
    for label in ('default', 'upsample x3', 'gat fp32', 'mdr fp32', 'upsample fp32', 'all fp32'):
        assert np.abs(outs[label] - outs['all x3']).max() * 1e3 <= 1.5e-3, label


def test_x3_split_is_exact(monkeypatch):
    """hi + mid + lo reproduces every fp32 value bit for bit (the premise of the scheme), checked through the vertex regressor:
    with one-hot activations the kernel must return single weights (+ bias + template) exactly."""
    monkeypatch.setenv('GATOR_UPSAMPLE_X3', '1')      # the three-plane regressor (the default two-plane one rounds to 22 bits)
    z, m = build_model('h36m17_bn', 'fused')
    sd = m.state_dict()
    w = sd['pose2mesh.upsample_conv.weight'].double().cpu()      # [6890, 431, 3]
    b = sd['pose2mesh.upsample_conv.bias'].double().cpu()
    vc = torch.zeros(32, 431, 3)
    for i in range(32):
        vc[i, (13 * i) % 431, i % 3] = 1.0
    base = m.pose2mesh.upsample(torch.zeros_like(vc).cuda()).double().cpu()      # bias + template
    out = m.pose2mesh.upsample(vc.cuda()).double().cpu()
    ref = torch.nn.functional.conv1d(vc.double(), w, b, padding=1) - b[None, :, None]
    got = out - base
    # (x + bias) + template in fp32 vs exact: only the final roundings differ -> compare against the fp32 evaluation of the same sum
    tol = 2 * np.finfo(np.float32).eps * float(base.abs().max() + ref.abs().max())
    assert float((got - ref).abs().max()) <= tol


def test_two_plane_regressor_error_and_scaling(monkeypatch):
    """The default vertex regressor carries each operand as two fp16 planes (upsample_x2.hip).  Against the exact product in
    float64: (a) realistic operands stay within fp32-arithmetic noise, (b) weights and activations far from 1 (2^-12 .. 2^6) lose
    nothing to fp16's range -- the pack-time power-of-two scaling keeps both planes normal -- and (c) the result does not depend
    on the batch it was computed in (one kernel shape for every batch size)."""
    monkeypatch.setenv('GATOR_UPSAMPLE_X3', '2')
    z, m = build_model('h36m17_bn', 'fused')
    sd = m.state_dict()
    w = sd['pose2mesh.upsample_conv.weight'].double().cpu()
    base = m.pose2mesh.upsample(torch.zeros(1, 431, 3).cuda()).double().cpu()      # bias + template
    g = torch.Generator().manual_seed(5)
    for B, mag in ((200, 0.3), (33, 2.0 ** -12), (5, 64.0)):
        vc = (torch.randn(B, 431, 3, generator=g) * mag).float()
        out = m.pose2mesh.upsample(vc.cuda()).double().cpu()
        ref = torch.nn.functional.conv1d(vc.double(), w, None, padding=1) + base
        scale = float(ref.abs().max())
        err = float((out - ref).abs().max())
        print('\n[two-plane regressor] B=%d |x|~%.1e: max err %.2e (outputs up to %.2e)' % (B, mag, err, scale))
        assert err <= 4 * np.finfo(np.float32).eps * scale + 1e-7 * mag
        one = m.pose2mesh.upsample(vc[:1].cuda()).cpu()
        assert torch.equal(one[0], m.pose2mesh.upsample(vc.cuda()).cpu()[0])
