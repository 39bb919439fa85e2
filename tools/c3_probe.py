"""Config-3 probe: accuracy (vs the fp64 oracle) and speed of gator_forward_bf16 beside the fp32 forward, one box.
usage: python tools/c3_probe.py [B ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup
from oracle import gator_oracle as go

def timed(fn, steps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    return sorted(ts)[2]

def main():
    Bs = [int(a) for a in sys.argv[1:]] or [256, 2048]
    for name in ('coco19_alpha', 'h36m17_bn'):
        z, m = build_model(name, 'fused')
        zz, c, sd = oracle_setup(name)
        n = 64
        x = torch.from_numpy(synthetic.synthetic_pose2d(max(Bs), c.J, seed=31))
        ref, rp = go.gator_forward(sd, c, x[:n], torch.float64)
        for B in Bs:
            xb = x[:B].cuda()
            for prec in ('f32', 'bf16'):
                m.precision = prec
                v, p3 = m(xb)
                torch.cuda.synchronize()
                e = np.abs(v[:n].cpu().numpy().astype(np.float64) - ref.numpy()[:min(n, B)]) * 1e3 if B >= n else None
                dt = timed(lambda: m(xb))
                st = m.device_status() if hasattr(m, 'device_status') else None
                print('%-13s B=%5d %-4s: %8.1f meshes/s  %.4f ms | err vs fp64 (first %d): max %.3e rms %.3e mm | status %s' % (
                    name, B, prec, B / dt, dt * 1e3, n, e.max() if e is not None else -1, np.sqrt((e ** 2).mean()) if e is not None else -1, st), flush=True)

main()
