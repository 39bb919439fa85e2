"""Config 3 (B=2048 J=19, 16-bit mode): libraries x MDR chunk sizes, one box.  usage: python tools/c3_chunk_probe.py lib1 lib2 ..."""
import os, subprocess, sys
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model
B = 2048
x = torch.from_numpy(synthetic.synthetic_pose2d(B, 19, seed=31)).cuda()
z, m = build_model('coco19_alpha', 'fused')
m.precision = sys.argv[1]
for _ in range(3): m(x)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 10)
print('%.4f' % (sorted(ts)[2] * 1e3))
'''
libs = sys.argv[1:] or ['default']
for rep in range(2):
    for lib in libs:
        row = []
        for chunk in ('f32', 0, 342, 512, 683, 1024, 2048):
            env = dict(os.environ)
            if lib != 'default': env['GATOR_AMD_LIB'] = os.path.abspath(lib)
            prec = 'bf16'
            if chunk == 'f32': prec = 'f32'
            elif chunk: env['GATOR_MDR_PERSIST_CHUNK'] = str(chunk)
            r = subprocess.run([sys.executable, '-c', CHILD, prec], env=env, capture_output=True, text=True)
            row.append('%s=%s' % (chunk if chunk else 'dflt256', r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'ERR'))
        print('%-28s rep %d  ms: %s' % (os.path.basename(lib), rep, '  '.join(row)), flush=True)
