"""The "next" rows either side of the path on the device, against the oracle:
input pipeline (pelvis/neck + standardisation) and the Procrustes alignment under PA-MPJPE."""
import numpy as np
import pytest
import torch

from gator_amd import eval as geval
from gator_amd import preprocess
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def test_demo_input_matches_reference_chain():
    """tests/golden/demo_preprocess.npz was produced by the REAL reference chain (add_pelvis/add_neck, get_bbox, process_bbox,
    j2d_processing, /[288,384], standardise) on demo/coco_joint_input.npy."""
    z = load_golden('demo_preprocess')
    raw = torch.from_numpy(z['raw_coco17'].astype(np.float32))[None].cuda()
    out = preprocess.coco_to_model_input(raw).cpu().numpy()
    assert out.shape == (1, 19, 2)
    assert np.abs(out - z['pose2d']).max() <= 2e-6


@pytest.mark.parametrize('B,J,C,add', [(1, 17, 3, True), (257, 17, 2, True), (64, 17, 2, False), (33, 19, 3, False)])
def test_preprocess_vs_oracle(B, J, C, add):
    from oracle import gator_oracle as go
    rs = np.random.RandomState(B + J)
    raw = (rs.rand(B, J, C) * np.array([640, 480, 1][:C]) + np.array([100, 50, 0][:C])).astype(np.float32)
    out = preprocess.normalise_pose2d(torch.from_numpy(raw).cuda(), add_pelvis_neck=add).cpu().numpy()
    for b in range(B):
        j = raw[b].astype(np.float64)
        if add:
            j3 = np.concatenate([j, np.ones((J, 3 - C))], 1) if C < 3 else j
            j = go.add_pelvis_neck_coco(j3)
        ref = go.normalise_pose2d(j)
        assert np.abs(out[b] - ref).max() <= 2e-6, b
    # per-sample per-axis zero mean / unit population std: the distribution the model is fed
    assert np.abs(out.mean(1)).max() < 1e-5 and np.abs(out.std(1) - 1).max() < 1e-5


def test_preprocess_rejects_host_and_bad_shapes():
    with pytest.raises(RuntimeError):
        preprocess.normalise_pose2d(torch.zeros(2, 17, 2))
    with pytest.raises(ValueError):
        preprocess.coco_to_model_input(torch.zeros(2, 19, 2).cuda())


@pytest.mark.parametrize('case', ['random', 'reflection', 'coplanar', 'scaled'])
def test_rigid_align_vs_oracle(case):
    from oracle import gator_oracle as go
    rs = np.random.RandomState(7)
    B, N = 65, 14
    a = rs.randn(B, N, 3) * 300.0
    Rm = np.linalg.qr(rs.randn(3, 3))[0]
    b = (a @ Rm.T) * 1.1 + rs.randn(B, 1, 3) * 50 + rs.randn(B, N, 3) * 20.0
    if case == 'reflection':
        b = b * np.array([1.0, 1.0, -1.0])              # best orthogonal map is a reflection: exercises the det < 0 fix
    if case == 'coplanar':
        a[:, :, 2] = 0.0                                # rank-2 covariance
        b = (a @ Rm.T) + rs.randn(B, N, 3) * 5.0
    if case == 'scaled':
        a, b = a * 1e-3, b * 1e-3                       # metres instead of millimetres
    got = geval.rigid_align(torch.from_numpy(a.astype(np.float32)).cuda(), torch.from_numpy(b.astype(np.float32)).cuda()).cpu().numpy()
    scale = np.abs(b).max()
    for i in range(B):
        ref = go.rigid_align(a[i].astype(np.float32), b[i].astype(np.float32))
        assert np.abs(got[i] - ref).max() <= 3e-7 * scale + 1e-6 * np.abs(ref).max(), (case, i)
    # PA-MPJPE end to end
    p = float(geval.pa_mpjpe(torch.from_numpy(a.astype(np.float32)).cuda(), torch.from_numpy(b.astype(np.float32)).cuda(), eval_joints=tuple(range(N))))
    r = go.pa_mpjpe(a.astype(np.float32), b.astype(np.float32))
    assert abs(p - r) <= 1e-5 * max(1.0, r)


def test_general_preprocess_chain_kernel_matches_reference_golden():
    """gator_preprocess_chain_f32 against the reference's own chain (rotation, flips, rejected boxes) and, for rot = 0 / no flip,
    against the reduced kernel."""
    from gator_amd import preprocess as pp
    from tests.helpers import load_golden
    z = load_golden('preprocess_chain')
    j = torch.from_numpy(z['joints'].astype(np.float32)).cuda()
    out, valid = pp.preprocess_chain(j, rot_deg=z['rot'].astype(np.float32), flip=z['flip'], flip_pairs=[tuple(p) for p in z['flip_pairs']])
    out, valid = out.cpu().numpy(), valid.cpu().numpy()
    assert np.array_equal(valid, z['valid'])
    ok = z['valid'] == 1
    err = np.abs(out[ok] - z['pose2d'][ok]).max()
    print('\ngeneral preprocess chain vs reference: max|d| %.2e over %d samples' % (err, int(ok.sum())))
    assert err < 2e-5                       # float32 inputs (the fixture's joints are float64) + float32 output
    assert np.all(out[~ok] == 0)
    plain, v2 = pp.preprocess_chain(j)
    red = pp.normalise_pose2d(j)
    assert np.abs(plain[ok].cpu().numpy() - red[ok].cpu().numpy()).max() < 5e-5


def test_emulated_gather_traffic_copies_the_shard():
    """gator_emulate_gather_traffic (include/gator_hip.h; tools/contention_model.py): `copies` images of the source behind each other, whatever the
    workgroup count; bad arguments are refused."""
    import ctypes
    from gator_amd import _lib
    lib = _lib.load()
    src = torch.randn(3 * 4096 + 4, device='cuda')
    dst = torch.zeros(5 * src.numel(), device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n_wg in (1, 16, 300):
        dst.zero_()
        _lib.check(lib.gator_emulate_gather_traffic(src.data_ptr(), dst.data_ptr(), src.numel() * 4, 5, n_wg, st), 'gator_emulate_gather_traffic')
        torch.cuda.synchronize()
        assert torch.equal(dst.view(5, -1), src.expand(5, -1))
    assert lib.gator_emulate_gather_traffic(src.data_ptr(), dst.data_ptr(), 10, 1, 1, st) == -1      # GATOR_EINVAL: not a multiple of 16
