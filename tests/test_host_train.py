"""Host-side pieces of the training row that need no GPU: the ctypes mirrors of the C structs, the vertex incidence lists that
order the face-loss gradient gathers, the learning-rate schedule, the parameter / buffer split of a reference state_dict."""
import ctypes

import numpy as np

from gator_amd import _lib, synthetic
from gator_amd.train import losses, model as M
from gator_amd.train.optim import Adam
from tests.helpers import golden_shapes, load_golden


def test_ctypes_structs_mirror_the_c_abi():
    lib = _lib.load()
    assert ctypes.sizeof(_lib.GemmProblem) == lib.gator_t_struct_size(0)
    assert lib.gator_t_struct_size(7) == -1


def test_vertex_incidence_lists():
    faces = synthetic.synthetic_faces(3, num_faces=500, num_verts=120)
    ptr, idx = losses.vertex_incidence(faces, 120)
    assert ptr[0] == 0 and ptr[-1] == 1500 and len(idx) == 1500
    flat = faces.reshape(-1)
    for v in (0, 7, 59, 119):
        mine = idx[ptr[v]:ptr[v + 1]]
        assert np.array_equal(np.sort(mine), np.nonzero(flat == v)[0]) and np.all(np.diff(mine) > 0)     # every (face, corner), ascending
    assert (np.diff(ptr) >= 1).all()                      # synthetic_faces uses every vertex
    f = synthetic.synthetic_faces(0)
    assert f.shape == (13776, 3) and (f[:, 0] != f[:, 1]).all() and (f[:, 0] != f[:, 2]).all() and (f[:, 1] != f[:, 2]).all()


def test_multistep_lr_follows_the_reference_loop():
    a = Adam.__new__(Adam)
    a.base_lr, a.gamma, a.milestones = 1e-3, 0.1, (30,)
    lrs = {}
    for e in (1, 2, 30, 31, 40):
        a.epoch = e
        lrs[e] = a.lr
    assert lrs[1] == lrs[30] == 1e-3 and abs(lrs[31] - 1e-4) < 1e-12 and abs(lrs[40] - 1e-4) < 1e-12     # main/train.py:36-39, config.py:76-77


def test_parameter_buffer_split_and_rates():
    z = load_golden('h36m17_bn')
    keys = list(golden_shapes(z))
    bufs = [k for k in keys if M.is_buffer(k)]
    assert sorted(b.rsplit('.', 1)[-1] for b in bufs) == sorted(['graph_adj', 'init_vertices', 'init_vertices', 'init_vertices_6890', 'running_mean',
                                                                  'running_var', 'num_batches_tracked'])
    names = [str(k) for k in load_golden('train_h36m17_bn')['param_names']]
    assert sorted(k for k in keys if not M.is_buffer(k)) == names          # exactly the reference's named_parameters()
    r = M.Rates()
    assert r.gat_path[0] == 0.0 and abs(r.gat_path[-1] - 0.2) < 1e-12 and r.gat_attn == 0.4 and r.mdr_self == 0.1
    assert M.Rates(0.0).gat_attn == 0.0
