"""Pin the oracle: it must reproduce what the REAL reference produced (tests/golden/*.npz, made by
tools/gen_golden.py from /root/reference) -- constants exactly, activations/vertices to fp32 noise."""
import numpy as np
import pytest
import torch

from oracle import gator_oracle as go
from oracle import graph_consts as gc
from tests.helpers import VARIANTS, oracle_setup, load_golden


@pytest.mark.parametrize('name', VARIANTS)
def test_constants_match_reference(name):
    z, c, sd = oracle_setup(name)
    assert np.array_equal(c.graph_adj, z['graph_adj'])
    assert np.array_equal(c.sp, z['shortest_path'])
    assert np.array_equal(c.path, z['path'])
    assert np.array_equal(c.vj, z['vj_relation'])
    assert np.array_equal(c.v431, z['init_vertices_431'])
    # template joints come from a float32 matmul whose summation order differs (numpy vs MKL): 1 ulp
    assert np.abs(c.edge_input - z['edge_input']).max() < 2e-7
    # seeded-weight recipe drift guard
    assert np.array_equal(sd['pose_lifter.lifter.weight'].numpy()[:2, :8], z['w_probe_lifter'])
    assert np.array_equal(sd['pose2mesh.upsample_conv.weight'].numpy()[:2, :3], z['w_probe_upconv'])


def test_graph_statistics():
    """SURVEY Appendix D2 (probed from the reference's build_adj + GAT.py:59-64 deletions)."""
    for J, nnz, D, deg in ((17, 49, 8, [4, 3, 3, 2, 3, 3, 2, 3, 5, 3, 2, 3, 3, 2, 3, 3, 2]),
                           (19, 71, 7, [4, 4, 4, 3, 3, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 3, 3, 4, 5])):
        sk, fl = gc.joint_setting(J)
        a = gc.delete_symmetric_edges(gc.build_adj(J, sk, fl))
        sp, path = gc.floyd_warshall(a)
        assert int(a.sum()) == nnz and int(sp.max()) == D
        assert a.astype(np.int64).sum(1).tolist() == deg
        assert (sp < 10).all() and (np.diag(sp) == 0).all()
        # direct neighbours have no intermediate node; expand(path) has length sp
        for i in range(J):
            for j in range(J):
                if i != j:
                    assert len(gc.get_all_edges(path, i, j)) == sp[i, j] - 1


@pytest.mark.parametrize('name', VARIANTS)
def test_forward_matches_reference(name):
    z, c, sd = oracle_setup(name)
    x = torch.from_numpy(z['pose2d'])
    taps = {}
    mesh64, p64 = go.gator_forward(sd, c, x, torch.float64, taps)
    # fp64 oracle == fp64 reference to ~1e-5 mm (only float32-stored constants differ in the last bit)
    assert np.abs(mesh64.numpy() - z['verts_f64']).max() * 1e3 < 5e-5
    assert np.abs(p64.numpy() - z['pose3d_f64']).max() < 1e-6
    for k in ('hop_path_bias', 'feat', 'gat_block0', 'gat_block5', 'mdr_lbf2', 'vert431'):
        ref = z[k].astype(np.float64)
        assert np.abs(taps[k].numpy() - ref).max() < 2.5e-6 * max(1.0, np.abs(ref).max()), k   # taps are the fp32 reference run
    mesh32, p32 = go.gator_forward(sd, c, x, torch.float32)
    # two fp32 evaluations (oracle / reference) agree to the reference's own fp32 noise (6e-4 mm)
    assert np.abs(mesh32.numpy().astype(np.float64) - z['verts_f64']).max() * 1e3 < 1e-3
    assert np.abs(z['verts'].astype(np.float64) - z['verts_f64']).max() * 1e3 < 1e-3


def test_demo_preprocess_plumbing():
    """BASELINE config 1: demo/coco_joint_input.npy -> [1,19,2] normalised input (demo/run.py:103-133)."""
    z = load_golden('demo_preprocess')
    j19 = go.add_pelvis_neck_coco(z['raw_coco17'])
    x = go.normalise_pose2d(j19)
    assert x.shape == (19, 2)
    assert np.abs(x - z['pose2d'][0]).max() < 5e-6       # float32 cast at lib/aug_utils.py:63
    assert abs(z['pose2d'].min() + 2.0051) < 1e-3 and abs(z['pose2d'].max() - 2.1714) < 1e-3


@pytest.mark.parametrize('fixture,gain', [('scale_gain02', 0.2), ('scale_gain10', 1.0)])
def test_scale_fixtures_pin_the_oracle(fixture, gain):
    """B=64 at two output-head gains (the REAL reference's fp32 run and its fp64 anchor on a fixed 512-vertex subset): the fp64
    oracle reproduces the anchor, and the fp32 oracle is as close to it as the reference's own fp32 run is (scale-free)."""
    z = load_golden(fixture)
    zz, c, sd = oracle_setup(str(z['variant']), upsample_gain=gain)
    x = torch.from_numpy(z['pose2d'])
    sub = z['vertex_subset'].astype(np.int64)
    mesh64, p64 = go.gator_forward(sd, c, x, torch.float64)
    scale = float(z['verts_absmax'])
    assert np.abs(mesh64.numpy()[:, sub] - z['verts_f64']).max() * 1e3 < 5e-5 * max(1.0, scale)
    assert np.abs(p64.numpy() - z['pose3d_f64']).max() < 1e-6
    ref_noise = float(np.abs(z['ref32_minus_f64'].astype(np.float64)).max() * 1e3)
    mesh32, _ = go.gator_forward(sd, c, x, torch.float32)
    ours = float(np.abs(mesh32.numpy()[:, sub].astype(np.float64) - z['verts_f64']).max() * 1e3)
    assert ours <= 1.5 * ref_noise, (ours, ref_noise)


def test_rigid_align_matches_reference():
    """oracle.rigid_align / pa_mpjpe == lib/coord_utils.py:127-149 run on random, mirrored, near-coplanar and scaled sets."""
    z = load_golden('rigid_align')
    out = np.stack([go.rigid_align(a, b) for a, b in zip(z['A'], z['B'])])
    assert np.abs(out - z['aligned']).max() < 1e-9 * np.abs(z['aligned']).max()
    assert abs(go.pa_mpjpe(z['A'], z['B']) - float(z['pa_mpjpe'])) < 1e-9


def test_general_preprocess_chain_matches_reference():
    """oracle.preprocess_pose2d == the reference's bbox/affine/flip/normalise chain run as it is on 48 random detections with
    rotations and flips (tests/golden/preprocess_chain.npz), including the boxes process_bbox rejects (None)."""
    z = load_golden('preprocess_chain')
    fp = [tuple(p) for p in z['flip_pairs']]
    for i in range(len(z['rot'])):
        o = go.preprocess_pose2d(z['joints'][i], z['rot'][i], bool(z['flip'][i]), fp)
        if z['valid'][i] == 0:
            assert o is None
        else:
            assert np.abs(o - z['pose2d'][i]).max() < 1e-6, i
    # rot = 0, no flip: the chain reduces to plain standardisation (what gator_preprocess_pose2d_f32 computes)
    i = int(np.flatnonzero((z['rot'] == 0) & (z['flip'] == 0) & (z['valid'] == 1))[0])
    assert np.abs(go.preprocess_pose2d(z['joints'][i]) - go.normalise_pose2d(z['joints'][i])).max() < 5e-6
