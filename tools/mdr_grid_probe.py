"""MDR stage time by persistent-grid size (workgroups per CU = waves per SIMD): how much does the second wave of a SIMD add?
usage: python tools/mdr_grid_probe.py   (one process per setting; GATOR_MDR_PERSIST_GRID is read at ctx creation)"""
import os, subprocess, sys
CHILD = r'''
import os, sys, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model
B, prec = int(sys.argv[1]), sys.argv[2]
x = torch.from_numpy(synthetic.synthetic_pose2d(B, 17, seed=31)).cuda()
z, m = build_model('h36m17_bn', 'fused')
m.precision = prec
m(x); torch.cuda.synchronize()
m.profile(1)
for _ in range(12): m(x)
torch.cuda.synchronize()
prof = m.profile_read()
print('%.4f' % (prof['mdr_layers'][0] / prof['mdr_layers'][1]))
'''
for prec in ('f32', 'bf16'):
    row = []
    for grid in (128, 256, 384, 512):
        env = dict(os.environ, GATOR_MDR_PERSIST='1', GATOR_MDR_PERSIST_GRID=str(grid))
        r = subprocess.run([sys.executable, '-c', CHILD, '256', prec], env=env, capture_output=True, text=True)
        row.append('%d WGs: %s ms' % (grid, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'ERR ' + r.stderr[-200:]))
    print('B=256 %-4s MDR launch by grid:  %s' % (prec, '   '.join(row)), flush=True)
