"""Digest of the forward's outputs for a few (variant, batch) cases: run it under two builds of the library (GATOR_AMD_LIB=...) and
compare the lines -- equal digests = bitwise equal results.  python tools/ab_digest.py [tag]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gator_amd import synthetic
from tests.helpers import build_model

def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('GATOR_AMD_LIB', 'default'))
    for name, J in (('h36m17_bn', 17), ('coco19_alpha', 19)):
        z, m = build_model(name, 'fused')
        for B in (5, 256, 700):
            x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=B)).cuda()
            v, p = m(x)
            torch.cuda.synchronize()
            h = hashlib.sha256(v.cpu().numpy().tobytes() + p.cpu().numpy().tobytes()).hexdigest()[:16]
            print('%-24s %-14s B=%-4d %s  finite=%s' % (tag, name, B, h, bool(torch.isfinite(v).all())), flush=True)

if __name__ == '__main__':
    main()
