"""Shared test plumbing: rebuild (state_dict, base data, oracle constants) of a golden variant."""
import ast
import os

import numpy as np
import torch

from gator_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
VARIANTS = ('h36m17_bn', 'coco19_alpha')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def golden_shapes(z):
    return {k: (ast.literal_eval(str(s)), str(k).endswith('num_batches_tracked'))
            for k, s in zip(z['state_dict_keys'], z['state_dict_shapes'])}


def variant_setup(name, upsample_gain=0.2, seed=None):
    """-> (z, J, alpha, base, numpy seeded weights {key: ndarray})  (same stream order as tools/gen_golden.py).
    `seed` overrides the golden's weight / base-data seed (other weight draws of the same recipe: tools/error_budget.py)."""
    z = load_golden(name)
    J, alpha = int(z['num_joint']), bool(z['alpha'])
    seed = int(z['seed']) if seed is None else int(seed)
    base = synthetic.make_base_data(seed)
    weights = synthetic.seeded_state_dict(golden_shapes(z), base['rs'], upsample_gain=upsample_gain)
    return z, J, alpha, base, weights


def oracle_setup(name, upsample_gain=0.2, seed=None):
    from oracle import gator_oracle as go
    z, J, alpha, base, weights = variant_setup(name, upsample_gain, seed)
    c = go.Consts(J, synthetic.model_j_regressor(J), base, alpha)
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    sd['pose_lifter.graph_adj'] = torch.from_numpy(c.graph_adj)
    return z, c, sd


def build_model(name, impl='fused', device='cuda', upsample_gain=0.2, seed=None):
    """gator_amd GATOR module of a golden variant with the seeded weights loaded, on `device`."""
    import scipy.sparse as sps
    from gator_amd import models
    from gator_amd.models.GAT import _dense_adj  # noqa: F401
    z, J, alpha, base, weights = variant_setup(name, upsample_gain, seed)
    sk, fl = _joint_setting(J)
    adj = np.zeros((J, J))
    for a, b in tuple(sk) + tuple(fl):
        adj[a, b] = adj[b, a] = 1
    graph_adj = [None, sps.csr_matrix(adj + np.eye(J))]
    m = models.GATOR.get_model(J, 128, 6, graph_adj, 1, torch.Tensor(synthetic.model_j_regressor(J)), base_data=base, alpha=alpha)
    sd = m.state_dict()
    sd.update({k: torch.from_numpy(v) for k, v in weights.items()})
    m.load_state_dict(sd)
    m.impl = impl
    if device is not None:
        m = m.to(device)
    m.eval()
    return z, m


def _joint_setting(J):
    # joint-set tables are caller-side data (data/Human36M/dataset.py:56-59,70-74 in the reference)
    if J == 17:
        return (((0, 7), (7, 8), (8, 9), (9, 10), (8, 11), (11, 12), (12, 13), (8, 14), (14, 15), (15, 16), (0, 1), (1, 2),
                 (2, 3), (0, 4), (4, 5), (5, 6)), ((1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13)))
    return (((1, 2), (0, 1), (0, 2), (2, 4), (1, 3), (6, 8), (8, 10), (5, 7), (7, 9), (12, 14), (14, 16), (11, 13), (13, 15),
             (17, 11), (17, 12), (17, 18), (18, 5), (18, 6), (18, 0)),
            ((1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)))
