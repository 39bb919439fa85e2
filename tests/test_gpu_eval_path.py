"""Rows either side of the hot path, on the GPU: BASELINE config 1 (demo plumbing: real COCO-17 sample -> J=19 forward),
SURVEY 8f-1 joint regression (lib/core/base.py:221) and the config-5 evaluation pipeline (forward -> x1000 -> J_regressor_h36m ->
root-align -> 14 eval joints -> MPJPE / PA-MPJPE, data/PW3D/dataset.py:273-286) against the oracle."""
import numpy as np
import pytest
import torch

from gator_amd import eval as geval
from gator_amd import synthetic
from tests.helpers import build_model, load_golden, oracle_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def coco():
    z, m = build_model('coco19_alpha', 'fused')
    zz, c, sd = oracle_setup('coco19_alpha')
    return m, c, sd


def test_demo_sample_forward(coco):
    """demo/coco_joint_input.npy -> add pelvis/neck -> normalise (checked vs the reference in the CPU suite) -> forward."""
    from oracle import gator_oracle as go
    m, c, sd = coco
    d = load_golden('demo_preprocess')
    x = torch.from_numpy(d['pose2d'])                                    # [1,19,2] produced by the reference's own pipeline
    ours = go.normalise_pose2d(go.add_pelvis_neck_coco(d['raw_coco17']))
    assert np.abs(ours - d['pose2d'][0]).max() < 5e-6
    ref, rp = go.gator_forward(sd, c, x, torch.float64)
    verts, pose3d = m(x.cuda())
    assert float(np.abs(verts.cpu().numpy() - ref.numpy()).max() * 1e3) <= 1e-3
    assert float(np.abs(pose3d.cpu().numpy() - rp.numpy()).max()) <= 1e-3


@pytest.mark.parametrize('which', ['h36m', 'coco'])
def test_joint_regression(which):
    jr = synthetic.load_j_regressors()[which]
    reg = geval.JointRegressor(jr, 'cuda')
    assert reg.nnz in (105, 107)
    rs = np.random.RandomState(3)
    for B in (1, 5, 64):
        v = rs.randn(B, 6890, 3).astype(np.float32)
        got = reg(torch.from_numpy(v).cuda()).cpu().numpy()
        ref = np.einsum('jv,bvc->bjc', jr.astype(np.float64), v.astype(np.float64))
        assert got.shape == (B, 17, 3)
        assert np.abs(got - ref).max() < 2e-6


def test_eval_pipeline_config5(coco):
    """Mixed 'gt-like' / 'det-like' synthetic 3DPW inputs: GPU metrics == oracle metrics."""
    from oracle import gator_oracle as go
    m, c, sd = coco
    B = 24
    clean = synthetic.synthetic_pose2d(B // 2, 19, seed=11)
    det = synthetic.synthetic_pose2d(B // 2, 19, seed=12, jitter=0.05)
    x = torch.from_numpy(np.concatenate([clean, det], 0))
    jr = synthetic.load_j_regressors()['h36m']
    # synthetic ground truth: the fp64 oracle mesh of a perturbed input (so errors are tens of mm, like a real eval)
    gt_mesh, _ = go.gator_forward(sd, c, torch.from_numpy(synthetic.synthetic_pose2d(B, 19, seed=13)), torch.float64)
    gt_mesh = gt_mesh * 1000
    gt_joint = go.regress_joints(jr, gt_mesh)
    ref_mesh, _ = go.gator_forward(sd, c, x, torch.float64)
    ref_mesh = ref_mesh * 1000
    ref_joint = go.regress_joints(jr, ref_mesh)
    e_ref = go.mpjpe(ref_joint.numpy(), gt_joint.numpy(), list(geval.H36M_EVAL_JOINTS))
    pa_ref = go.pa_mpjpe(ref_joint.numpy(), gt_joint.numpy(), geval.H36M_EVAL_JOINTS)
    verts, _ = m(x.cuda())
    mesh = verts * 1000                                                        # lib/core/base.py:219
    joints = geval.JointRegressor(jr, 'cuda')(mesh)
    gtj = gt_joint.float().cuda()
    e = float(geval.mpjpe(joints, gtj))
    pa = float(geval.pa_mpjpe(joints, gtj))
    ev = float(geval.mpvpe(mesh, gt_mesh.float().cuda(), joints, gtj))
    print('\n[config5] MPJPE %.4f mm (oracle %.4f)  PA-MPJPE %.4f mm (oracle %.4f)  MPVPE %.3f mm' % (e, e_ref, pa, pa_ref, ev))
    assert abs(e - e_ref) < 1e-3 and abs(pa - pa_ref) < 1e-3
    assert e > 1.0          # a real, non-degenerate error
