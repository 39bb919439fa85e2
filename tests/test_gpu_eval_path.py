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


@pytest.mark.parametrize('name,which', [('coco19_alpha', 'h36m'), ('h36m17_bn', 'h36m'), ('coco19_alpha', 'coco')])
def test_fused_joint_regression_epilogue(name, which):
    """gator_forward_joints_f32: the regressor's partial products are formed in the vertex GEMM's epilogue and summed in a fixed
    order -- same joints as regressing the materialised mesh (and as the fp64 oracle), vertices (when asked for) bit-identical
    to the plain forward, and identical joints whether or not the vertices are stored."""
    from oracle import gator_oracle as go
    z, m = build_model(name, 'fused')
    zz, c, sd = oracle_setup(name)
    jr = synthetic.load_j_regressors()[which]
    m.set_joint_regressor(jr)
    for B in (1, 33, 70):
        x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=40 + B))
        verts, pose3d = m(x.cuda())
        j1, p1 = m.forward_joints(x.cuda())
        j2, p2, v2 = m.forward_joints(x.cuda(), with_verts=True)
        assert torch.equal(v2, verts) and torch.equal(p1, pose3d) and torch.equal(p2, pose3d)
        assert torch.equal(j1, j2)
        sep = geval.JointRegressor(jr, 'cuda')(verts)
        assert float((j1 - sep).abs().max()) < 2e-6                       # metres; both sum fp32 products in fp64
        ref, _ = go.gator_forward(sd, c, x, torch.float64)
        rj = go.regress_joints(jr, ref).numpy()
        assert np.abs(j1.cpu().numpy() - rj).max() * 1e3 <= 1e-3         # 1e-3 mm against the fp64 oracle


def test_eval_mode_b2048_without_vertices():
    """BASELINE config 5 at B=2048 (J=19 model, mixed gt-/det-like inputs): ShardedForward(mode='eval') regresses the joints in
    the vertex GEMM's epilogue and reduces MPJPE / PA-MPJPE sums on the device -- no mesh is ever stored or moved.  Against the
    same metrics from materialised meshes (all 2048) and the numpy oracle (a 32-sample slice)."""
    from oracle import gator_oracle as go
    from gator_amd.parallel import ShardedForward
    B = 2048
    z, m = build_model('coco19_alpha', 'fused')
    zz, c, sd = oracle_setup('coco19_alpha')
    clean = synthetic.synthetic_pose2d(B // 2, 19, seed=21)
    det = synthetic.synthetic_pose2d(B // 2, 19, seed=22, jitter=0.05)
    x = torch.from_numpy(np.concatenate([clean, det], 0)).cuda()
    jr = synthetic.load_j_regressors()['h36m']
    tgt = torch.from_numpy(np.random.RandomState(5).randn(B, 17, 3).astype(np.float32) * 120).cuda()
    run = ShardedForward(m, 1, 0, None, micro_batch=1024, mode='eval')
    run.set_eval(jr, tgt)
    got = run.step(x)
    torch.cuda.synchronize()
    assert float(got[2]) == B
    verts, _ = m(x)
    joints = geval.JointRegressor(jr, 'cuda')(verts) * 1000.0
    want = torch.stack([geval.mpjpe(joints, tgt) * B, geval.pa_mpjpe(joints, tgt) * B]).double()
    print('\n[config5 B=2048] MPJPE %.4f mm, PA-MPJPE %.4f mm (fused epilogue) vs %.4f / %.4f (materialised meshes)'
          % (float(got[0]) / B, float(got[1]) / B, float(want[0]) / B, float(want[1]) / B))
    assert torch.allclose(got[:2], want, rtol=2e-6)
    sl = slice(1000, 1032)                                                  # straddles nothing special; oracle check on a slice
    ref, _ = go.gator_forward(sd, c, x[sl].cpu(), torch.float64)
    rj = go.regress_joints(jr, ref * 1000).numpy()
    j_f, _ = m.forward_joints(x[sl].contiguous())
    e_ref = go.mpjpe(rj, tgt[sl].cpu().numpy().astype(np.float64), list(geval.H36M_EVAL_JOINTS))
    e_got = float(geval.mpjpe(j_f * 1000.0, tgt[sl]))
    assert abs(e_ref - e_got) < 1e-3


def test_fused_joint_errors_kernel():
    """gator_joint_errors_f32 (per-sample MPJPE and PA-MPJPE in one launch) == the numpy oracle (data/PW3D/dataset.py:273-286 and the
    per-sample Procrustes loop :337-375) and == the separate device functions, incl. the reference's rigid_align golden sets."""
    from oracle import gator_oracle as go
    from tests.helpers import load_golden
    rs = np.random.RandomState(8)
    B = 70
    pred = (rs.randn(B, 17, 3) * 0.2).astype(np.float32)                   # metres
    tgt = (pred * 1000 + rs.randn(B, 17, 3) * 40).astype(np.float32)        # mm
    pred[3] = pred[3][:, [0, 1, 2]] * np.array([1, 1, 1e-4], np.float32)    # nearly coplanar
    err = geval.joint_errors(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), pred_scale=1000.0).cpu().numpy().astype(np.float64)
    ev = list(geval.H36M_EVAL_JOINTS)
    for b in (0, 3, 17, 69):
        p64 = (pred[b] * np.float32(1000)).astype(np.float64)
        want0 = go.mpjpe(p64[None], tgt[b][None].astype(np.float64), ev)
        pr, tr = p64 - p64[0:1], tgt[b].astype(np.float64) - tgt[b][0:1].astype(np.float64)
        want1 = go.pa_mpjpe(pr[None], tr[None], ev)
        assert abs(err[b, 0] - want0) <= 2e-5 * max(1.0, want0) and abs(err[b, 1] - want1) <= 2e-5 * max(1.0, want1), (b, err[b], want0, want1)
    mm = torch.from_numpy(pred).cuda() * 1000.0
    assert abs(float(geval.mpjpe(mm, torch.from_numpy(tgt).cuda())) - err[:, 0].mean()) < 1e-3
    assert abs(float(geval.pa_mpjpe(mm, torch.from_numpy(tgt).cuda())) - err[:, 1].mean()) < 1e-3
    z = load_golden('rigid_align')                                          # the reference's own alignments: all 14 points, root 0
    a, b = z['A'].astype(np.float32), z['B'].astype(np.float32)
    e = geval.joint_errors(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), eval_joints=None).cpu().numpy()
    want = np.array([np.sqrt((((o - o[0]) * 0 + (go.rigid_align(x - x[0], y - y[0]) - (y - y[0]))) ** 2).sum(1)).mean() for x, y, o in zip(z['A'], z['B'], z['aligned'])])
    assert np.abs(e[:, 1] - want).max() <= 1e-4 * want.max()
