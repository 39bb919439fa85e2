"""Error budget of the split-precision stages: max / rms vertex error (mm) against the fp64 oracle with each stage's
X3 switch on or off (GATOR_GAT_X3, GATOR_MDR_X3, GATOR_UPSAMPLE_X3 are read when the context is created)."""
import os, sys, json, subprocess
import numpy as np

CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup
from oracle import gator_oracle as go
out = {}
for name, B, seed in [('h36m17_bn', 64, 5), ('coco19_alpha', 64, 6), ('h36m17_bn', 64, 7)]:
    z, m = build_model(name, 'fused')
    zz, c, sd = oracle_setup(name)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, c.J, seed=seed))
    ref, rp = go.gator_forward(sd, c, x, torch.float64)
    v, p = m(x.cuda())
    e = np.abs(v.cpu().numpy().astype(np.float64) - ref.numpy()) * 1e3
    out['%s/%d' % (name, seed)] = [float(e.max()), float(np.sqrt((e ** 2).mean()))]
print(json.dumps(out))
'''

def main():
    for gat, mdr, up in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)]:
        env = dict(os.environ, GATOR_GAT_X3=str(gat), GATOR_MDR_X3=str(mdr), GATOR_UPSAMPLE_X3=str(up))
        r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        print('gat=%d mdr=%d up=%d' % (gat, mdr, up), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)

if __name__ == '__main__':
    main()
