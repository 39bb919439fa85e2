"""Is the forward's clock held down under load?  The same binary and batch on the seeded weights / poses and on all-zero weights and
poses (same instruction stream up to the softmax rescale branch; the data toggles nothing): per the MI355X guide's 'DVFS give-back'
a zero-data run holds a higher clock, so the ratio of the two times is the clock the real data gives up.  python tools/dvfs_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gator_amd import synthetic
from tests.helpers import build_model


def timed(m, x, out, n=200):
    for _ in range(20):
        m(x, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            m(x, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


def main():
    B, J = 256, 17
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=1)).cuda()
    out = (torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, J, 3, device='cuda'))
    z, m = build_model('h36m17_bn', 'fused')
    t_real = timed(m, x, out)
    z0, m0 = build_model('h36m17_bn', 'fused')
    with torch.no_grad():
        for p in m0.parameters():
            p.zero_()
        for b in m0.buffers():
            if b.dtype.is_floating_point:
                b.zero_()
    x0 = torch.zeros_like(x)
    t_zero = timed(m0, x0, out)
    t_real2 = timed(m, x, out)
    print('B=256 forward: seeded data %.1f us | all-zero weights and poses %.1f us | seeded again %.1f us  ->  zero / real = %.3f'
          % (t_real, t_zero, t_real2, t_zero / (0.5 * (t_real + t_real2))))


if __name__ == '__main__':
    main()
