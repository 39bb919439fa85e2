import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup
from oracle import gator_oracle as go
for name in ('h36m17_bn', 'coco19_alpha'):
    zz, c, sd = oracle_setup(name)
    for B in (1, 7, 8, 40):
        x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=11 + B))
        ref, rp = go.gator_forward(sd, c, x, torch.float64)
        os.environ['GATOR_GAT_TILED'] = '1'
        z, m = build_model(name, 'fused')
        v, p = m(x.cuda()); torch.cuda.synchronize()
        os.environ['GATOR_GAT_TILED'] = '0'
        z, m0 = build_model(name, 'fused')
        v0, p0 = m0(x.cuda()); torch.cuda.synchronize()
        e = np.abs(v.cpu().numpy() - ref.numpy()).max() * 1e3
        e0 = np.abs(v0.cpu().numpy() - ref.numpy()).max() * 1e3
        ep = np.abs(p.cpu().numpy() - rp.numpy()).max()
        print('%s B=%d tiled: verts %.3e mm pose3d %.3e mm | S=1 kernel: %.3e mm | tiled vs S=1 %.3e mm' % (name, B, e, ep, e0, (v - v0).abs().max().item() * 1e3), flush=True)
