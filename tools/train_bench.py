#!/usr/bin/env python3
"""Training-row measurement: one step = forward in training mode (dropout on) + five losses + backward + Adam of the
full GATOR model on B synthetic samples resident in HBM (reference batch size 64, lib/core/config.py:69).
Prints ONE JSON line: samples/s eager and as a replayed hipGraph, and the oracle's torch-CPU autograd step beside it.
  python tools/train_bench.py [--batch 64] [--joints 17] [--steps 20] [--warmup 3] [--no-graph] [--no-cpu-baseline]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--joints', type=int, default=17)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-eager', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()
    from gator_amd import synthetic
    from gator_amd.train.trainer import Trainer
    from tests.helpers import build_model
    name = 'h36m17_bn' if a.joints == 17 else 'coco19_alpha'
    z, m = build_model(name, 'fused')
    seed = int(z['seed'])
    base = synthetic.make_base_data(seed)
    jreg = synthetic.load_j_regressors()['h36m'].astype(np.float32)
    faces = synthetic.synthetic_faces(seed)
    x = torch.from_numpy(synthetic.synthetic_pose2d(a.batch, a.joints, 3)).cuda()
    tg = {k: torch.from_numpy(v).cuda() for k, v in synthetic.training_targets(a.batch, a.joints, base, jreg, 3).items()}

    def timed(tr):
        for _ in range(a.warmup):
            tr.step(x, tg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss, _ = tr.step(x, tg)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, float(loss)

    out = {'metric': 'training samples/sec (forward + losses + backward + Adam, dropout on)', 'unit': 'samples/s', 'batch': a.batch, 'joints': a.joints,
           'steps': a.steps, 'warmup': a.warmup, 'dtype': 'f32', 'data': 'synthetic', 'n_params': None}
    if not a.no_eager:
        tr = Trainer.from_module(m, faces, jreg, seed=1)
        tr.epoch = 16
        out['n_params'] = int(sum(b - a_ for a_, b, _ in tr.params.slots))
        ms, loss = timed(tr)
        out['eager'] = {'ms_per_step': round(ms, 3), 'samples_per_s': round(a.batch / ms * 1e3, 1), 'last_loss': loss}
    if not a.no_graph:
        tr = Trainer.from_module(m, faces, jreg, seed=1)
        tr.epoch = 16
        tr.capture(x, tg)
        ms, loss = timed(tr)
        out['graph'] = {'ms_per_step': round(ms, 3), 'samples_per_s': round(a.batch / ms * 1e3, 1), 'last_loss': loss}
        out['value'] = out['graph']['samples_per_s']
        # algorithmic work of a step: 3 x the forward's 4.10e8 / 4.18e8 FLOP per sample (backward = 2 x forward), priced
        # against the fp32-input MFMA peak every product of the step runs on
        flop = 3.0 * (4.10e8 if a.joints == 17 else 4.18e8)
        out['roofline'] = {'bound': 'mfma', 'achieved': round(out['value'] * flop / 1e12, 2), 'peak': 157.3, 'unit': 'TFLOP/s',
                           'frac': round(out['value'] * flop / 157.3e12, 4), 'flop_per_sample': flop, 'note': 'whole step, algorithmic FLOPs; launch-bound at B=64'}
    if not a.no_cpu_baseline:
        from oracle import gator_oracle as go
        from gator_amd.train.model import is_buffer
        from tests.helpers import oracle_setup
        zz, c, sd = oracle_setup(name)

        def cpu_step(P, leaves, xb, tgb):
            mesh, p3 = go.gator_forward_train(P, c, xb, torch.float32)
            loss, _ = go.training_loss(mesh, p3, tgb, jreg, faces, with_edge=True)
            torch.autograd.grad(loss, leaves, allow_unused=True)

        def cpu_batch(B):
            xb = torch.from_numpy(synthetic.synthetic_pose2d(B, a.joints, 3))
            tgb = {k: torch.from_numpy(v) for k, v in synthetic.training_targets(B, a.joints, base, jreg, 3).items()}
            return xb, tgb

        # BASELINE.md section 3's protocol, as bench.py applies it to the forward: thread count by a short probe (8 / 16 / 32: all host
        # threads is pathological for these small tensors), then B in {16, 64, 256}, warm-up + timed steps, median per B, best B
        P = {k: (v.float().requires_grad_(True) if (v.is_floating_point() and not is_buffer(k)) else v) for k, v in sd.items()}
        leaves = [v for v in P.values() if torch.is_tensor(v) and v.requires_grad]
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cands = sorted({n for n in (8, 16, 32) if n <= avail}) or [avail]
        xb, tgb = cpu_batch(64)
        probe = {}
        for nt in cands:
            torch.set_num_threads(nt)
            cpu_step(P, leaves, xb, tgb)
            t0 = time.perf_counter()
            cpu_step(P, leaves, xb, tgb)
            probe[nt] = time.perf_counter() - t0
        nt = min(probe, key=probe.get)
        torch.set_num_threads(nt)
        per_b = {}
        for B, (nw, ntimed) in ((16, (3, 10)), (64, (3, 10)), (256, (1, 3))):
            xb, tgb = cpu_batch(B)
            for _ in range(nw):
                cpu_step(P, leaves, xb, tgb)
            ts = []
            for _ in range(ntimed):
                t0 = time.perf_counter()
                cpu_step(P, leaves, xb, tgb)
                ts.append(time.perf_counter() - t0)
            per_b[B] = B / float(np.median(ts))
        bb = max(per_b, key=per_b.get)
        out['cpu_baseline'] = {'value': round(per_b[bb], 1), 'unit': 'samples/s', 'cores': int(nt), 'kind': 'port',
                               'sample': 'oracle forward(train, no dropout) + five losses + torch-CPU autograd backward, no optimizer; B in {16,64,256} x '
                                         '(3 warm-up + 10 timed; B=256: 1 + 3), median per B, best B=%d; %d of %d host threads (probe over %s)' % (bb, nt, avail, cands),
                               'per_batch': {str(k): round(v, 1) for k, v in per_b.items()}}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
