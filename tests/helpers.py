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


def variant_setup(name):
    """-> (z, J, alpha, base, numpy seeded weights {key: ndarray})  (same stream order as tools/gen_golden.py)."""
    z = load_golden(name)
    J, alpha, seed = int(z['num_joint']), bool(z['alpha']), int(z['seed'])
    base = synthetic.make_base_data(seed)
    weights = synthetic.seeded_state_dict(golden_shapes(z), base['rs'])
    return z, J, alpha, base, weights


def oracle_setup(name):
    from oracle import gator_oracle as go
    z, J, alpha, base, weights = variant_setup(name)
    c = go.Consts(J, synthetic.model_j_regressor(J), base, alpha)
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    sd['pose_lifter.graph_adj'] = torch.from_numpy(c.graph_adj)
    return z, c, sd
