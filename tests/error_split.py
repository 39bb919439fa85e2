"""Where the device forward's fp32 noise is made (test infrastructure; companion of tests/error_budget.py).

    python -m tests.error_split [--samples 512] [--variant h36m17_bn]

Three runs against the fp64 oracle on the same poses, rms / max vertex error in mm:
  full     the device forward;
  gat      only the ENCODER on the device: its pose3d and feat (fp32) are fed to the oracle's fp64 MDR;
  mdr      only the MDR head + vertex regressor on the device: it is fed the oracle's fp64 pose_combine (rounded to fp32);
and the same three for the reference's own arithmetic (the oracle in fp32).  gat^2 + mdr^2 ~ full^2 when the parts are independent."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=512)
    ap.add_argument('--variant', default='h36m17_bn')
    ap.add_argument('--seed', type=int, default=None)
    a = ap.parse_args()
    import torch
    from gator_amd import synthetic
    from oracle import gator_oracle as go
    from tests.helpers import build_model, oracle_setup
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    z, m = build_model(a.variant, 'fused', seed=a.seed)
    zz, c, sd = oracle_setup(a.variant, seed=a.seed)
    N, J = a.samples, c.J
    x = torch.from_numpy(synthetic.synthetic_pose2d(N, J, seed=4242))

    def oracle(gdt, mdt, gat_out=None):
        with torch.no_grad():
            p2 = x.to(gdt)
            if gat_out is None:
                xo, feat = go.gat_forward(sd, c, p2.reshape(N, -1), gdt)
            else:
                xo, feat = gat_out
            p3 = xo.reshape(-1, J, 3)
            pc = torch.cat((x.to(mdt), p3.to(mdt) / 1000, feat.to(mdt)), 2)
            return go.mdr_forward(sd, c, pc, mdt).double(), pc

    ref, pc64 = oracle(torch.float64, torch.float64)
    rep = lambda tag, v: print('%-34s rms %.3e  max %.3e mm' % (tag, float(np.sqrt(((v - ref) ** 2).mean()) * 1e3), float((v - ref).abs().max() * 1e3)), flush=True)
    xd = x.cuda()
    v, p = m(xd)
    rep('device  full', v.cpu().double())
    xo, feat = m.pose_lifter(xd)
    rep('device  gat only (-> fp64 MDR)', oracle(torch.float64, torch.float64, (xo.cpu().double(), feat.cpu().double()))[0])
    rep('device  mdr only (<- fp64 GAT)', m.pose2mesh(pc64.float().cuda()).cpu().double())
    rep('ref32   full', oracle(torch.float32, torch.float32)[0])
    rep('ref32   gat only (-> fp64 MDR)', oracle(torch.float32, torch.float64)[0])
    rep('ref32   mdr only (<- fp64 GAT)', oracle(torch.float64, torch.float32)[0])
    # the encoder's own outputs
    xo64, f64 = go.gat_forward(sd, c, x.double().reshape(N, -1), torch.float64)
    print('device  pose3d max |d| %.3e mm (scale %.0f), feat max |d| %.3e rms %.3e (scale %.2f)' % (
        float((xo.cpu().double() - xo64).abs().max()), float(xo64.abs().max()), float((feat.cpu().double() - f64).abs().max()),
        float(((feat.cpu().double() - f64) ** 2).mean().sqrt()), float(f64.abs().max())))
    xo32, f32 = go.gat_forward(sd, c, x.reshape(N, -1), torch.float32)
    print('ref32   pose3d max |d| %.3e mm, feat max |d| %.3e rms %.3e' % (float((xo32.double() - xo64).abs().max()), float((f32.double() - f64).abs().max()),
                                                                         float(((f32.double() - f64) ** 2).mean().sqrt())))


if __name__ == '__main__':
    main()
