#!/usr/bin/env python3
"""Is the oracle (bench.py `cpu_baseline.kind = "port"`) a fair stand-in for the reference's own CPU path?  

Dev-container tool: imports the REAL reference from /root/reference with tools/gen_golden.py's shims and times
`model(x)` (eval, no_grad, fp32) against `oracle.gator_oracle.gator_forward` on the same weights, the same poses, the same
thread count; B in {16, 64, 256}, 3 warm-up + `--reps` timed forwards each, median.  Each side runs in its OWN process, the
processes alternate (reference, oracle, reference, ... `--rounds` each) and the medians of the rounds are compared: timed in one
process the two disturb each other (the reference module keeps its three 380 MB attention maps alive, vanilla_transformer_encoder.py:91;
whichever runs second at B = 256 is up to 2 x slower -- measured both ways round).  One more process checks that the two agree
(<= 1e-3 mm) on the timed inputs.

    python tools/cpu_port_vs_reference.py [--reps 10] [--threads N] --json profiles/r06_cpu_port_vs_reference.json > profiles/r06_cpu_port_vs_reference.txt

The reference never travels to the GPU box; this figure (oracle / reference meshes per second) is what bench.py quotes beside its
CPU baseline as `port_vs_reference`.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import gen_golden as gg                      # noqa: E402  (the shims and the base-data writer)
from gator_amd import synthetic              # noqa: E402
from oracle import gator_oracle as go        # noqa: E402
from oracle import graph_consts as gc        # noqa: E402


def build_reference(J, alpha, seed):
    import scipy.sparse as sps
    scratch = tempfile.mkdtemp(prefix='gator_cpuport_')
    for m in [k for k in sys.modules if k.split('.')[0] in ('models', 'graph_utils', 'coarsening', 'core', 'funcs_utils')]:
        del sys.modules[m]
    gg.install_shims(scratch, alpha)
    base = synthetic.make_base_data(seed)
    skeleton, flips = gc.joint_setting(J)
    adj0 = gc.build_adj(J, skeleton, flips)
    sp, path = gc.floyd_warshall(gc.delete_symmetric_edges(adj0))
    gg.write_base_data(scratch, base, sp, path, '3dpw' if J == 19 else 'h36m')
    os.chdir(scratch)
    import models  # noqa  (reference lib/models)
    from models.backbones import mesh as ref_mesh
    ref_mesh.Mesh.__init__.__defaults__ = ('data/base_data/mesh_downsampling.npz', 1, 1, torch.device('cpu'))
    model = models.GATOR.get_model(J, 128, 6, [None, sps.csr_matrix(adj0)], 1, torch.Tensor(synthetic.model_j_regressor(J)))
    model.eval()
    sd = model.state_dict()
    new = synthetic.seeded_state_dict(synthetic.shapes_of(sd), base['rs'], upsample_gain=0.2)
    sd.update({k: torch.from_numpy(v) for k, v in new.items()})
    model.load_state_dict(sd)
    c = go.Consts(J, synthetic.model_j_regressor(J), base, alpha)
    osd = {k: v.clone() for k, v in model.state_dict().items()}
    osd['pose_lifter.graph_adj'] = torch.from_numpy(c.graph_adj)
    os.chdir(REPO)
    return model, scratch, c, osd


def worker(side, J, reps, nt):
    torch.set_num_threads(nt)
    go.KEEP_ATTENTION_MAPS = True        # the reference's modules keep theirs (vanilla_transformer_encoder.py:91)
    alpha = J == 19
    model, scratch, c, osd = build_reference(J, alpha, seed=0 if J == 17 else 100)
    res = {}
    for B in (16, 64, 256):
        x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=1))
        f = (lambda: model(x)) if side == 'reference' else (lambda: go.gator_forward(osd, c, x, torch.float32))
        with torch.no_grad():
            if side == 'check':
                vr, _ = model(x)
                vo, _ = go.gator_forward(osd, c, x, torch.float32)
                res[str(B)] = float((vr - vo).abs().max()) * 1e3
                continue
            for _ in range(3):
                f()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        res[str(B)] = float(np.median(ts))
    print('RESULT ' + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--joints', type=int, default=17)
    ap.add_argument('--json', default='')
    ap.add_argument('--side', default='')
    a = ap.parse_args()
    nt = a.threads or (os.cpu_count() or 1)
    if a.side:
        return worker(a.side, a.joints, a.reps, nt)
    import subprocess

    def run(side):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--side', side, '--reps', str(a.reps), '--threads', str(nt), '--joints', str(a.joints)],
                           capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
        if not line:
            sys.stderr.write(r.stderr[-2000:])
            raise RuntimeError('worker %s failed' % side)
        return json.loads(line[-1][7:])

    print('reference (imported from /root/reference, shimmed) vs oracle/gator_oracle.py: fp32, eval, no_grad, %d threads, J=%d; per process 3 warm-up + median of %d; '
          '%d processes per side, alternating; median over the processes' % (nt, a.joints, a.reps, a.rounds))
    agree = run('check')
    tr, to = [], []
    for _ in range(a.rounds):
        tr.append(run('reference'))
        to.append(run('oracle'))
    out = {'threads': nt, 'joints': a.joints, 'reps': a.reps, 'rounds': a.rounds, 'per_batch': {}}
    for B in (16, 64, 256):
        mr = float(np.median([t[str(B)] for t in tr]))
        mo = float(np.median([t[str(B)] for t in to]))
        out['per_batch'][str(B)] = {'reference_meshes_per_s': round(B / mr, 1), 'oracle_meshes_per_s': round(B / mo, 1), 'oracle_over_reference': round(mr / mo, 3),
                                    'agreement_mm': agree[str(B)], 'reference_s_by_process': [round(t[str(B)], 4) for t in tr],
                                    'oracle_s_by_process': [round(t[str(B)], 4) for t in to]}
        print('B=%3d  reference %8.1f meshes/s (%s s)   oracle %8.1f meshes/s (%s s)   oracle / reference = %.3f   max |difference| %.2e mm'
              % (B, B / mr, ' '.join('%.3f' % t[str(B)] for t in tr), B / mo, ' '.join('%.3f' % t[str(B)] for t in to), mr / mo, agree[str(B)]), flush=True)
    best_r = max(v['reference_meshes_per_s'] for v in out['per_batch'].values())
    best_o = max(v['oracle_meshes_per_s'] for v in out['per_batch'].values())
    out['best_B_ratio'] = round(best_o / best_r, 3)
    print('best-B figures (what bench.py reports): reference %.1f, oracle %.1f meshes/s -> oracle / reference = %.3f' % (best_r, best_o, best_o / best_r))
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
