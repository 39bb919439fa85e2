"""Digest of the forward's outputs for a few (variant, batch, encoder pin) cases.
  * A/B of two builds: run it under each (GATOR_AMD_LIB=...) and compare the lines -- equal digests = bitwise equal results.
  * `--write tests/golden/fp32_digests.json` records the digests of the CURRENT library (on a GPU box); tests/test_gpu_digest.py then
    holds every later build to them, so that a change that is meant to be a schedule only (round 4: dead token rows, C-layout GELU,
    zero operands on dead lanes) is checked bit for bit by the suite instead of by hand.  Re-record only with a change that is MEANT to
    move bits, together with the error budget (tests/error_budget.py).
python tools/ab_digest.py [tag] [--write path]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gator_amd import synthetic
from tests.helpers import build_model

CASES = [(name, J, B, pin) for name, J in (('h36m17_bn', 17), ('coco19_alpha', 19)) for B, pin in ((5, 'auto'), (256, 'auto'), (700, 'auto'), (700, 'tiled'))]


def digests():
    out = {}
    models = {}
    for name, J, B, pin in CASES:
        if name not in models:
            models[name] = build_model(name, 'fused')[1]
        m = models[name]
        m.set_encoder(pin)
        x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=B)).cuda()
        v, p = m(x)
        torch.cuda.synchronize()
        m.set_encoder('auto')
        assert bool(torch.isfinite(v).all())
        out['%s B=%d %s' % (name, B, pin)] = hashlib.sha256(v.cpu().numpy().tobytes() + p.cpu().numpy().tobytes()).hexdigest()[:32]
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    write = sys.argv[sys.argv.index('--write') + 1] if '--write' in sys.argv else None
    if write in args:
        args.remove(write)
    tag = args[0] if args else os.path.basename(os.environ.get('GATOR_AMD_LIB', 'default'))
    d = digests()
    for k, h in d.items():
        print('%-24s %-32s %s' % (tag, k, h), flush=True)
    if write:
        with open(write, 'w') as f:
            json.dump({'what': 'sha256[:32] of (vertices, pose3d) bytes of gator_forward_f32, default arithmetic, gfx950; inputs synthetic_pose2d(B, J, seed=B), golden weights',
                       'digests': d}, f, indent=1)


if __name__ == '__main__':
    main()
