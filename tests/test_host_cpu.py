"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol include/gator_hip.h declares,
the host-side graph helpers match the oracle, and the lib/models mirror has the reference's state_dict layout."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from gator_amd import _lib
from oracle import graph_consts as gc
from tests.helpers import VARIANTS, build_model, golden_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = ''.join(open(os.path.join(ROOT, 'include', h)).read() for h in ('gator_hip.h', 'gator_train.h'))
    declared = set(re.findall(r'\b(gator_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in _lib.load().gator_version()


@pytest.mark.parametrize('J', [17, 19])
def test_host_graph_helpers_match_oracle(J):
    lib = _lib.load()
    sk, fl = gc.joint_setting(J)
    adj = gc.delete_symmetric_edges(gc.build_adj(J, sk, fl))
    sp, path = np.zeros((J, J), np.int64), np.zeros((J, J), np.int64)
    assert lib.gator_floyd_warshall(adj.ctypes.data, J, sp.ctypes.data, path.ctypes.data) == 0
    osp, opath = gc.floyd_warshall(adj)
    assert np.array_equal(sp, osp) and np.array_equal(path, opath)
    rs = np.random.RandomState(J)
    ed = np.triu(rs.rand(J, J).astype(np.float32), 1) * (adj > 0)
    D = int(sp.max())
    out = np.zeros((J, J, D), np.float32)
    assert lib.gator_gen_edge_input(path.ctypes.data, ed.ctypes.data, J, D, out.ctypes.data) == 0
    assert np.array_equal(out, gc.gen_edg_input(D, opath, ed))
    joints, verts = rs.randn(17, 3).astype(np.float32), rs.randn(431, 3).astype(np.float32)
    rel = np.zeros(431, np.int32)
    assert lib.gator_verts_joints_relation(joints.ctypes.data, 17, verts.ctypes.data, 431, rel.ctypes.data) == 0
    assert np.array_equal(rel, gc.build_verts_joints_relation(joints, verts))
    # disconnected graph: unreachable pairs carry the 510 sentinel in both outputs
    iso = np.eye(4, dtype=np.float32)
    iso[0, 1] = iso[1, 0] = 1
    sp4, p4 = np.zeros((4, 4), np.int64), np.zeros((4, 4), np.int64)
    assert lib.gator_floyd_warshall(iso.ctypes.data, 4, sp4.ctypes.data, p4.ctypes.data) == 0
    assert sp4[0, 2] == 510 and p4[0, 2] == 510 and sp4[0, 1] == 1
    assert lib.gator_floyd_warshall(None, 4, sp4.ctypes.data, p4.ctypes.data) != 0
    assert b'bad arguments' in lib.gator_last_error()


@pytest.mark.parametrize('name', VARIANTS)
def test_state_dict_layout_matches_reference(name):
    z, m = build_model(name, device=None)
    sd = m.state_dict()
    gs = golden_shapes(z)
    assert set(sd) == set(gs)
    for k, (shape, is_int) in gs.items():
        assert tuple(sd[k].shape) == shape, k
        assert (sd[k].dtype == torch.int64) == is_int, k
    assert np.array_equal(sd['pose_lifter.graph_adj'].numpy(), z['graph_adj'])
    assert np.array_equal(m.pose_lifter.spatial_pos, z['shortest_path'])
    assert np.array_equal(m.pose_lifter.path, z['path'])
    assert np.array_equal(m.pose_lifter.edge_input, z['edge_input'])
    assert np.array_equal(m.pose2mesh.vj_relation, z['vj_relation'])
    assert np.array_equal(sd['pose2mesh.init_vertices'].numpy(), z['init_vertices_431'])
    # a checkpoint dict in the reference's layout (main/train.py:51-58) round-trips
    ck = {'epoch': 0, 'model_state_dict': sd}
    m.load_state_dict(ck['model_state_dict'])
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, int(z['num_joint']), 2))      # no CPU path: must fail loudly


def test_bad_configuration_is_rejected():
    from gator_amd import models
    with pytest.raises(ValueError):
        models.GAT.get_model(17, 256, 4, graph_adj=[np.eye(17)], J_regressor=np.zeros((17, 6890), np.float32), base_data={})
    with pytest.raises(ValueError):
        models.GAT.get_model(24, 128, 6, graph_adj=[np.eye(24)], J_regressor=np.zeros((24, 6890), np.float32), base_data={})


def test_gat8_roles_keep_the_same_barrier_sequence():
    """csrc/gat_roles.hip: the product waves and the helper waves of k_gat8 run disjoint branches of one kernel that meet at
    numbered workgroup barriers; a barrier that only one role executes would hang the GPU.  Both branches must name exactly the
    sequence 0, 1 .. 22, each number once and in order."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gator_amd', 'csrc', 'gat_roles.hip')).read()
    body = src[src.index('void k_gat8('):src.index('__global__ void k_gather_tiles')]
    body = re.sub(r'#ifdef GATOR_DIAG.*?#endif', '', body, flags=re.S)       # (the diagnostic library's stamps / experiments)
    head, rest = body.split('product waves ====', 1)
    product, helper = rest.split('helper waves ====', 1)
    assert 'GAT8_BAR' not in head                              # the embedding uses plain __syncthreads() in uniform code
    want = list(range(23))
    for role, text in (('product', product), ('helper', helper)):
        got = [int(n) for n in re.findall(r'GAT8_BAR\((\d+)\)', text)]
        assert got == want, (role, got)
    # and neither role leaves its branch early: one `return` (the product waves', after their last barrier)
    assert product.count('return;') == 1 and helper.count('return;') == 0
    # round 6: the fused tail (gat8_tail, its own barrier inside) is entered exactly once by each role, after the numbered sequence
    for role, text in (('product', product), ('helper', helper)):
        assert len(re.findall(r'gat8_tail<', text)) == 1, role
        assert text.rindex('gat8_tail<') > text.rindex('GAT8_BAR(22)'), role
    tail = src[src.index('void gat8_tail('):src.index('// H4: the token-wise products on four partial products')]
    assert tail.count('__syncthreads()') == 1 and 'GAT8_BAR' not in tail
    assert tail.index('if (!joint) return;') < tail.index('__syncthreads()')      # the lifter-only exit is workgroup-uniform and ahead of the barrier
