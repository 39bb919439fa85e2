#!/usr/bin/env python3
"""Soak of the default forward on one GPU: N forwards with batch sizes drawn from 1..1500 (both encoder kernels, every queue shape of the persistent MDR launch,
chunked launches), each checked for finiteness and for bitwise agreement with the forward of the same inputs the first time that batch size was seen; every 50th also
against the four-launch MDR form with the two tail launches (GATOR_MDR_PERSIST=0, GATOR_GAT8_TAIL=0).  Prints one line.
    python tools/soak.py [N=5000] [seed=0]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gator_amd import synthetic               # noqa: E402
from tests.helpers import build_model         # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    z, m = build_model('h36m17_bn', 'fused')
    os.environ['GATOR_MDR_PERSIST'] = '0'
    os.environ['GATOR_GAT8_TAIL'] = '0'
    z0, m0 = build_model('h36m17_bn', 'fused')
    g = torch.Generator().manual_seed(seed)
    first, xs = {}, {}
    t0 = time.time()
    bad = 0
    for it in range(n):
        B = int(torch.randint(1, 1501, (1,), generator=g)) if it % 3 else int(torch.randint(1, 300, (1,), generator=g))
        if B not in xs:
            xs[B] = torch.from_numpy(synthetic.synthetic_pose2d(B, 17, seed=B)).cuda()
        v, p = m(xs[B])
        if B not in first:
            first[B] = (v.clone(), p.clone())
        else:
            if not (torch.equal(v, first[B][0]) and torch.equal(p, first[B][1])):
                bad += 1
                print('iteration %d B=%d: differs from the first forward of this batch size' % (it, B), flush=True)
        if it % 50 == 0:
            v0, p0 = m0(xs[B])
            if not (bool(torch.isfinite(v).all()) and torch.equal(v, v0) and torch.equal(p, p0)):
                bad += 1
                print('iteration %d B=%d: differs from the four-launch form' % (it, B), flush=True)
        if len(first) > 400:
            first.clear(); xs.clear()
    torch.cuda.synchronize()
    m.device_status()
    print('soak: %d forwards, %d mismatches, %.1f s' % (n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
