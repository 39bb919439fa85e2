"""BASELINE config 3: GATOR forward with the vertex regressor on bf16 MFMA (gator_forward_bf16).  Parity is MPJPE-level by
construction (bf16 rounding of the 431x3 coarse vertices and of upsample_conv.weight, fp32 accumulation): vertices within a
few mm (rms ~0.6 mm at these synthetic scales), regressed joints within 3 mm and MPJPE within 0.25 mm of the fp64 oracle; the fp32 stages are untouched."""
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
    assert err.max() < 8.0 and np.sqrt((err ** 2).mean()) < 1.0      # bf16 operand rounding: ~2^-9 relative per product
    assert np.abs(pose3d.cpu().numpy() - rp.numpy()).max() <= 1e-3          # GAT stays fp32
    jr = synthetic.load_j_regressors()['h36m']
    reg = geval.JointRegressor(jr, 'cuda')
    j_bf = reg(verts * 1000).cpu().numpy()
    j_ref = go.regress_joints(jr, ref * 1000).numpy()
    assert np.abs(j_bf - j_ref).max() < 3.0                                  # joints average ~6 vertices each
    gt = j_ref + np.random.RandomState(0).randn(*j_ref.shape) * 30.0
    e_bf = go.mpjpe(j_bf, gt, list(geval.H36M_EVAL_JOINTS))
    e_ref = go.mpjpe(j_ref, gt, list(geval.H36M_EVAL_JOINTS))
    print('[%s bf16] joints max %.3f mm, MPJPE %.4f vs %.4f mm' % (name, np.abs(j_bf - j_ref).max(), e_bf, e_ref))
    assert abs(e_bf - e_ref) < 0.25


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
