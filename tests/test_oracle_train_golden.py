"""Pins the oracle's TRAINING restatement (oracle/gator_oracle.py: gator_forward_train, training_loss) to the real reference:
loss parts and per-parameter gradients of one training step recorded by tools/gen_golden.py::train_golden (reference modules in
.train() with dropout p = 0, lib/core/loss.py criteria, lib/core/base.py:137-148 weighting, torch autograd)."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from gator_amd.train.model import is_buffer
from oracle import gator_oracle as go
from tests.helpers import load_golden, oracle_setup


def oracle_step(name, dtype=torch.float64, batch=None, seed_shift=0):
    z = load_golden('train_' + name)
    zz, c, sd = oracle_setup(name)
    J, seed = int(z['num_joint']), int(z['seed'])
    B = int(z['batch']) if batch is None else batch
    base = synthetic.make_base_data(seed)
    P = {k: (v.to(dtype).requires_grad_(True) if (v.is_floating_point() and not is_buffer(k)) else v) for k, v in sd.items()}
    pose2d = torch.from_numpy(z['pose2d'] if batch is None else synthetic.synthetic_pose2d(B, J, seed + 3 + seed_shift))
    jreg = synthetic.load_j_regressors()['h36m'].astype(np.float32)
    tg = {k: torch.from_numpy(v) for k, v in synthetic.training_targets(B, J, base, jreg, seed + seed_shift).items()}
    mesh, pose3d = go.gator_forward_train(P, c, pose2d, dtype)
    loss, parts = go.training_loss(mesh, pose3d, tg, jreg, synthetic.synthetic_faces(seed), with_edge=True)
    names = [k for k in P if torch.is_tensor(P[k]) and P[k].requires_grad]
    grads = dict(zip(names, torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)))
    return z, loss, parts, grads, (pose2d, tg, jreg)


@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
def test_oracle_training_step_matches_reference(name):
    z, loss, parts, grads, _ = oracle_step(name)
    want = z['loss_parts_f64']                                            # vertice, normal, edge, mesh2joint3d, liftedjoint3d, total
    got = [float(parts[k].detach()) for k in ('vertice', 'normal', 'edge', 'mesh2joint3d', 'liftedjoint3d')] + [float(loss.detach())]
    assert np.allclose(got, want, rtol=5e-9, atol=0)      # (the reference's fp64 run keeps a few float32 constants)
    names = [str(k) for k in z['param_names']]
    assert sorted(grads) == names                                         # the same parameter set as model.named_parameters()
    worst = 0.0
    for i, k in enumerate(names):
        idx = z['probe_idx'][i]
        n = int((idx >= 0).sum())
        g = grads[k]
        g = torch.zeros_like(g) if g is None else g
        got = g.reshape(-1).numpy()[idx[:n]]
        scale = max(float(z['grad_absmax'][i]), 1e-300)
        err = np.abs(got - z['grad_f64'][i][:n]).max()
        assert abs(float(g.abs().max()) - float(z['grad_absmax'][i])) <= 1e-7 * scale + 1e-16, k
        assert err <= 1e-7 * scale + 1e-16, (k, err, scale)
        worst = max(worst, err / scale)
    print('\n[%s] oracle training step vs reference fp64: %d tensors, worst probe error / max|g| = %.1e' % (name, len(names), worst))
