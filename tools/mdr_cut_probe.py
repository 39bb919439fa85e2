"""Diagnostic library (GATOR_AMD_LIB=gator_amd/lib/libgator_hip_diag.so): MDR stage time with the weight / K-V L2 traffic cut (GATOR_MDR_CUT).
One process per cut (the switch is read once).  usage: python tools/mdr_cut_probe.py B prec"""
import os, sys, time
import torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model
B, prec = int(sys.argv[1]), sys.argv[2]
J = 19
x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=31)).cuda()
z, m = build_model('coco19_alpha', 'fused')
m.precision = prec
m(x); torch.cuda.synchronize()
m.profile(1)
for _ in range(12): m(x)
torch.cuda.synchronize()
prof = m.profile_read()
print('cut=%s B=%d %s: %s' % (os.environ.get('GATOR_MDR_CUT', '0'), B, prec, {k: round(v[0] / v[1], 4) for k, v in prof.items()}), flush=True)
