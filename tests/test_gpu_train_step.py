"""The training row end to end on the GPU: loss and per-parameter gradients of one training step from the HIP
kernels against (a) the recording of the REAL reference's step (tests/golden/train_*.npz, reference in .train() with dropout
p = 0) and (b) the oracle's torch-CPU float64 autograd at another batch size; then Adam against torch.optim.Adam, dropout
reproducibility, and trained weights going back onto the inference kernels."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from gator_amd.train import model as M
from gator_amd.train import ops
from gator_amd.train.trainer import Trainer
from tests.helpers import build_model, load_golden

pytestmark = pytest.mark.gpu


def make_trainer(name, rates=None, seed=0, **kw):
    zz, m = build_model(name, 'fused')
    z = load_golden('train_' + name)
    faces = synthetic.synthetic_faces(int(z['seed']))
    jreg = synthetic.load_j_regressors()['h36m'].astype(np.float32)
    return z, m, Trainer.from_module(m, faces, jreg, rates=rates if rates is not None else M.Rates(0.0), seed=seed, **kw), jreg


def batch_of(z, B=None, shift=0):
    J, seed = int(z['num_joint']), int(z['seed'])
    base = synthetic.make_base_data(seed)
    jreg = synthetic.load_j_regressors()['h36m'].astype(np.float32)
    if B is None:
        pose2d, B = z['pose2d'], int(z['batch'])
    else:
        pose2d = synthetic.synthetic_pose2d(B, J, seed + 3 + shift)
    tg = synthetic.training_targets(B, J, base, jreg, seed + shift)
    return torch.from_numpy(pose2d).cuda(), {k: torch.from_numpy(v).cuda() for k, v in tg.items()}


@pytest.mark.parametrize('name', ['h36m17_bn', 'coco19_alpha'])
def test_training_step_matches_reference_recording(name):
    """Scale-free criterion per parameter tensor: |ours - ref fp64| <= 4 x |ref fp32 - ref fp64| + 2e-5 max|g|."""
    z, m, tr, _ = make_trainer(name)
    tr.epoch = 16                                                    # > edge_loss_start: all five losses (base.py:145-147)
    x, tg = batch_of(z)
    loss, parts, grad = tr.loss_and_grad(x, tg)
    want = z['loss_parts_f64']
    got = [float(parts[k]) for k in ('vertice', 'normal', 'edge', 'mesh2joint3d', 'liftedjoint3d')] + [float(loss)]
    print('\n[%s] loss parts ours %s\n%s reference fp64 %s' % (name, np.round(got, 6).tolist(), ' ' * len(name), np.round(want, 6).tolist()))
    assert np.allclose(got, want, rtol=2e-5)
    g = grad.cpu().double().numpy()
    names = [str(k) for k in z['param_names']]
    assert sorted(tr.params.names) == names
    slot = dict(zip(tr.params.names, tr.params.slots))
    worst, worst_k = 0.0, None
    absmax = dict(zip(names, z['grad_absmax']))
    for i, k in enumerate(names):
        a, b, shape = slot[k]
        idx = z['probe_idx'][i]
        n = int((idx >= 0).sum())
        got = g[a:b][idx[:n]]
        scale, noise = float(z['grad_absmax'][i]), float(z['ref32_minus_f64_max'][i])
        err = np.abs(got - z['grad_f64'][i][:n]).max()
        # (the key bias of a softmax attention has an exactly-zero gradient: its noise floor is set by its weight's gradient)
        sibling = float(absmax.get(k[:-4] + 'weight', 0.0)) if k.endswith('.bias') else 0.0
        tol = 4.0 * noise + 2e-5 * scale + 3e-6 * sibling + 1e-12
        assert err <= tol, '%s: err %.3e tol %.3e (max|g| %.3e, ref fp32 noise %.3e)' % (k, err, tol, scale, noise)
        assert abs(np.abs(g[a:b]).max() - scale) <= 4.0 * noise + 1e-4 * scale + 3e-6 * sibling + 1e-12, k
        if scale > 1e-9 and err / scale > worst:
            worst, worst_k = err / scale, k
    print('[%s] %d parameter tensors; worst probe error / max|g| = %.2e (%s)' % (name, len(names), worst, worst_k))


def test_training_step_matches_oracle_autograd_other_batch():
    """B = 6 (not the recorded batch), J = 19 alpha variant, against torch-CPU float64 autograd of the oracle."""
    from tests.test_oracle_train_golden import oracle_step
    name = 'coco19_alpha'
    z, m, tr, _ = make_trainer(name)
    tr.epoch = 16
    x, tg = batch_of(z, 6, shift=9)
    loss, parts, grad = tr.loss_and_grad(x, tg)
    _, oloss, oparts, ograds, _ = oracle_step(name, batch=6, seed_shift=9)
    assert abs(float(loss) - float(oloss.detach())) <= 2e-5 * abs(float(oloss.detach()))
    g = grad.cpu().double()
    for k, (a, b, shape) in zip(tr.params.names, tr.params.slots):
        og = ograds[k]
        og = torch.zeros(shape, dtype=torch.float64) if og is None else og
        scale = float(og.abs().max())
        sib = ograds.get(k[:-4] + 'weight') if k.endswith('.bias') else None          # exactly-zero gradients: see above
        sibling = float(sib.abs().max()) if sib is not None else 0.0
        err = float((g[a:b].view(shape) - og).abs().max())
        assert err <= 1e-4 * scale + 3e-6 * sibling + 1e-9, '%s: %.3e vs max|g| %.3e' % (k, err, scale)


def test_adam_matches_torch_and_weights_return_to_inference_kernels():
    name = 'h36m17_bn'
    z, m, tr, _ = make_trainer(name, lr=2e-6)                          # Adam's first steps move every weight by ~lr: keep it first-order
    x, tg = batch_of(z, 8, shift=2)
    before = tr.params.flat.detach().clone()
    ref_p = before.cpu().clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=2e-6)
    losses = []
    for it in range(3):
        loss, parts, grad = tr.loss_and_grad(x, tg)
        tr.optim.step(grad)
        ref_p.grad = grad.cpu().clone()
        opt.step()
        losses.append(float(loss))
    d = (tr.params.flat.detach().cpu() - ref_p.detach()).abs().max()
    assert float(d) <= 2.4e-7, float(d)  # (an ulp of the largest weights)                                   # same update rule as torch.optim.Adam
    assert float((tr.params.flat.detach() - before).abs().max()) > 4e-6
    assert losses[2] < losses[0]                                        # three steps on one batch reduce its loss
    rm = tr.params.buffers['pose2mesh.bias_norm.running_mean']
    assert float(rm.abs().max()) > 0                                    # BatchNorm running statistics moved (momentum 0.1)
    m.load_state_dict(tr.state_dict())                                  # trained weights -> the fused inference kernels
    m.eval()
    verts, pose3d = m(x)
    P = {k: v for k, v in zip(tr.params.names, [tr.params.flat.detach()[a:b].view(s) for a, b, s in tr.params.slots])}
    mesh_eval, p3 = M.gator_forward(P, tr.consts, x, training=False, buffers=tr.params.buffers)
    assert float((verts - mesh_eval).abs().max()) * 1e3 <= 2e-3        # eval forward of the training path == inference path (mm)
    assert float((pose3d - p3).abs().max()) <= 2e-3


def test_dropout_training_step_is_reproducible_and_active():
    name = 'coco19_alpha'
    z, m, tr, _ = make_trainer(name, rates=M.Rates(), seed=5)
    x, tg = batch_of(z, 4, shift=1)
    l1, _, g1 = tr.loss_and_grad(x, tg)
    l2, _, g2 = tr.loss_and_grad(x, tg)                                  # the generator advanced: other masks
    z2, m2, tr2, _ = make_trainer(name, rates=M.Rates(), seed=5)
    l3, _, g3 = tr2.loss_and_grad(x, tg)                                 # fresh trainer, same seed: bit-identical
    assert torch.equal(g1, g3) and float(l1) == float(l3)
    assert not torch.equal(g1, g2)
    z0, m0, tr0, _ = make_trainer(name)
    l0, _, g0 = tr0.loss_and_grad(x, tg)
    assert float((g1 - g0).abs().max()) > 0
    assert torch.isfinite(g1).all()


def test_training_step_runs_only_library_kernels():
    """torch is plumbing in the training row: over one whole step (forward, losses, backward, Adam) the only aten operators that
    run are allocation / view metadata - every kernel on the data path comes from libgator_hip.so."""
    from torch.profiler import ProfilerActivity, profile
    z, m, tr, _ = make_trainer('h36m17_bn', rates=M.Rates(), seed=3)
    tr.epoch = 16
    x, tg = batch_of(z, 8, shift=4)
    tr.step(x, tg)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        tr.step(x, tg)
        torch.cuda.synchronize()
    metadata = {'aten::empty', 'aten::view', 'aten::as_strided', 'aten::empty_like', 'aten::empty_strided', 'aten::reshape', 'aten::expand',
                'aten::permute', 'aten::transpose', 'aten::narrow', 'aten::slice', 'aten::select', 'aten::detach', 'aten::alias',
                'aten::_reshape_alias', 'aten::unsqueeze', 'aten::t', 'aten::view_as', 'aten::_unsafe_view', 'aten::squeeze', 'aten::result_type'}
    foreign = sorted({e.key for e in prof.key_averages() if e.key.startswith('aten::') and e.key not in metadata})
    assert foreign == [], foreign


def test_captured_step_replays_like_eager_and_draws_new_masks():
    """Trainer.capture: the hipGraph replay of forward + losses + backward, with the optimiser outside, follows the eager step
    (dropout off: same weights after three steps to rounding); with dropout on, every replay draws different masks because the
    Philox offset's high word is read from the device step counter the graph itself advances."""
    name = 'h36m17_bn'
    z, m, eager, _ = make_trainer(name, lr=1e-5)
    z, m, graph, _ = make_trainer(name, lr=1e-5)
    x, tg = batch_of(z, 8, shift=6)
    graph.capture(x, tg)
    for it in range(3):
        x2, tg2 = batch_of(z, 8, shift=6 + it)                        # new data every step through the static input buffers
        le, _ = eager.step(x2, tg2)
        lg, _ = graph.step(x2, tg2)
        assert abs(float(le) - float(lg)) <= 1e-5 * abs(float(le)), it
    d = float((eager.params.flat.detach() - graph.params.flat.detach()).abs().max())
    assert d <= 2e-5, d                                               # Adam's sign-like first steps amplify rounding: a few lr
    z, m, drop, _ = make_trainer(name, rates=M.Rates(), seed=11, lr=0.0)
    drop.capture(x, tg)
    losses = [float(drop.step(x, tg)[0]) for _ in range(4)]           # lr = 0: only the masks change between replays
    assert len(set(losses)) == 4, losses
    assert all(np.isfinite(losses))


def test_captured_step_crosses_edge_loss_start_and_keeps_training_state():
    """lib/core/base.py:141-143: EdgeLengthLoss joins the loss from the epoch after `edge_loss_start` on.  A step replayed from a
    hipGraph must follow: the graph captured before the switch is not the graph replayed after it.  And capturing is not a
    training step: after eager steps (or optim.load_state_dict) the captured trainer continues with Adam's step index, and the
    BatchNorm running statistics are what the eager trainer holds (the warm-up passes of capture() leave no trace)."""
    name = 'h36m17_bn'
    z, m, eager, _ = make_trainer(name, lr=1e-5)
    z, m, graph, _ = make_trainer(name, lr=1e-5)
    for tr in (eager, graph):
        tr.edge_loss_start = 1
        tr.epoch = 1
    x, tg = batch_of(z, 8, shift=3)
    for it in range(2):                                                 # two eager steps on both, THEN capture one of them
        eager.step(x, tg)
        graph.step(x, tg)
    graph.capture(x, tg)
    assert graph.optim.step_count == 2 and int(graph.gen.counter.item()) == 2
    for k, v in eager.params.buffers.items():                           # running_mean / running_var / num_batches_tracked untouched by capture
        assert torch.equal(v, graph.params.buffers[k]), k
    for epoch in (1, 1, 2, 2, 1):                                       # crosses edge_loss_start forth and back under replay
        eager.epoch = graph.epoch = epoch
        le, pe = eager.step(x, tg)
        lg, pg = graph.step(x, tg)
        assert ('edge' in pg) == (epoch > 1) == ('edge' in pe), (epoch, sorted(pg))
        assert abs(float(le) - float(lg)) <= 1e-5 * abs(float(le)), (epoch, float(le), float(lg))
    assert graph.optim.step_count == eager.optim.step_count == 7 and int(graph.gen.counter.item()) == 7
    d = float((eager.params.flat.detach() - graph.params.flat.detach()).abs().max())
    assert d <= 4e-5, d                                                 # same bias correction on both: a wrong t = 1 would move weights by ~lr per step
    for k, v in eager.params.buffers.items():
        if v.is_floating_point():
            assert float((v - graph.params.buffers[k]).abs().max()) <= 1e-5 * (1.0 + float(v.abs().max())), k
        else:
            assert torch.equal(v, graph.params.buffers[k]), k
    with pytest.raises(ValueError):                                     # another batch shape needs a new capture
        graph.step(x[:4], {k: v[:4] for k, v in tg.items()})


def test_lift_trainer_matches_oracle_autograd():
    """LiftTrainer (lib/core/base.py:260-300): GAT alone, CoordLoss on the lifted joints - gradients against torch-CPU float64
    autograd over the oracle's gat_forward."""
    from gator_amd.train.trainer import LiftTrainer
    from oracle import gator_oracle as go
    from tests.helpers import oracle_setup
    name, B = 'coco19_alpha', 5
    zz, m = build_model(name, 'fused')
    lt = LiftTrainer.from_module(m.pose_lifter, rates=M.Rates(0.0))
    J = 19
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, 21))
    cam = torch.from_numpy(np.random.RandomState(3).randn(B, J, 3).astype(np.float32) * 250)
    valid = torch.ones(B, J, 1)
    valid[2] = 0
    loss, grad = lt.loss_and_grad(x.cuda().reshape(B, -1), cam.cuda(), valid.cuda())
    z, c, sd = oracle_setup(name)
    P = {k: (v.double().requires_grad_(True) if (v.is_floating_point() and not M.is_buffer(k)) else v) for k, v in sd.items() if k.startswith('pose_lifter.')}
    x_out, _ = go.gat_forward(P, c, x.reshape(B, -1), torch.float64)
    oloss = go.coord_loss(x_out.reshape(B, J, 3), cam.double(), valid.double())
    assert abs(float(loss) - float(oloss.detach())) <= 1e-5 * float(oloss.detach())
    g = grad.cpu().double()
    for k, (a, b, shape) in zip(lt.params.names, lt.params.slots):
        og, = torch.autograd.grad(oloss, P['pose_lifter.' + k], retain_graph=True, allow_unused=True)
        og = torch.zeros(shape, dtype=torch.float64) if og is None else og
        scale = float(og.abs().max())
        err = float((g[a:b].view(shape) - og).abs().max())
        assert err <= 1e-4 * scale + 1e-9, (k, err, scale)
    before = lt.params.flat.detach().clone()
    lt.step(x.cuda().reshape(B, -1), cam.cuda(), valid.cuda())
    assert float((lt.params.flat.detach() - before).abs().max()) > 0


def test_optimizer_state_interchanges_with_torch_adam():
    """main/train.py:51-58 / lib/core/base.py:73-77: the checkpoint's optim_state_dict is torch.optim.Adam's.  Ours loads into a
    torch Adam over the same parameters (and back) and both continue identically."""
    z, m, tr, _ = make_trainer('h36m17_bn', lr=1e-4)
    x, tg = batch_of(z, 4, shift=8)
    for _ in range(2):
        tr.step(x, tg)
    sd = tr.optim.state_dict()
    cpu_params = [tr.params.flat.detach()[a:b].view(s).cpu().clone().requires_grad_(True) for a, b, s in tr.params.slots]
    opt = torch.optim.Adam(cpu_params, lr=1e-4)
    opt.load_state_dict(sd)                                             # torch accepts our layout
    _, _, grad = tr.loss_and_grad(x, tg)
    for p, (a, b, s) in zip(cpu_params, tr.params.slots):
        p.grad = grad[a:b].view(s).cpu().clone()
    opt.step()
    tr.optim.step(grad)
    worst = max(float((p.detach() - tr.params.flat.detach()[a:b].view(s).cpu()).abs().max()) for p, (a, b, s) in zip(cpu_params, tr.params.slots))
    assert worst <= 2.4e-7, worst
    z2, m2, tr2, _ = make_trainer('h36m17_bn', lr=1e-4)
    tr2.optim.load_state_dict(opt.state_dict())                         # and torch's state loads into ours
    assert tr2.optim.step_count == 3
    assert float((tr2.optim.exp_avg - tr.optim.exp_avg).abs().max()) <= 1e-6 * float(tr.optim.exp_avg.abs().max())   # CPU lerp vs ours: ulps
    assert float((tr2.optim.exp_avg_sq - tr.optim.exp_avg_sq).abs().max()) <= 1e-6 * float(tr.optim.exp_avg_sq.abs().max())


def test_training_overfits_a_fixed_batch():
    """End to end: 40 captured steps (hipGraph replay, Adam lr 1e-4, dropout off) on one fixed batch drive its loss down - the
    vertex and joint terms by more than half - and everything stays finite; then the same with the reference's dropout rates."""
    name = 'coco19_alpha'
    z, m, tr, _ = make_trainer(name, lr=1e-4)
    tr.epoch = 16
    x, tg = batch_of(z, 16, shift=12)
    tr.capture(x, tg)
    first = last = None
    for it in range(40):
        loss, parts = tr.step(x, tg)
        if it == 0:
            first = {k: float(v) for k, v in parts.items()}
        last = {k: float(v) for k, v in parts.items()}
    print('\noverfit: first', {k: round(v, 4) for k, v in first.items()}, '\n         last ', {k: round(v, 4) for k, v in last.items()})
    assert all(np.isfinite(list(last.values())))
    assert last['vertice'] < 0.5 * first['vertice'] and last['liftedjoint3d'] <= first['liftedjoint3d']       # (random +-300 mm targets at weight 1e-3 move slowly)
    assert sum(last.values()) < 0.7 * sum(first.values())
    assert torch.isfinite(tr.params.flat).all()
    z, m, tr2, _ = make_trainer(name, rates=M.Rates(), seed=4, lr=1e-4)
    tr2.epoch = 16
    l0 = float(tr2.step(x, tg)[0])
    for it in range(25):
        l1 = float(tr2.step(x, tg)[0])
    assert np.isfinite(l1) and l1 < l0
