"""Config 3 (B=2048 J=19): what each part of the 16-bit mode contributes -- speed and error of (MDR only) / (+ encoder) / (+ regressor weights one plane) / all, one box.
usage: python tools/c3_parts_probe.py"""
import os, subprocess, sys
CHILD = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup
from oracle import gator_oracle as go
B, n = 2048, 128
x = torch.from_numpy(synthetic.synthetic_pose2d(B, 19, seed=31))
z, m = build_model('coco19_alpha', 'fused')
zz, c, sd = oracle_setup('coco19_alpha')
m.precision = sys.argv[1]
xc = x.cuda()
v, p = m(xc)
torch.cuda.synchronize()
ref, rp = go.gator_forward(sd, c, x[:n], torch.float64)
e = np.abs(v[:n].cpu().numpy().astype(np.float64) - ref.numpy()) * 1e3
ep = np.abs(p[:n].cpu().numpy().astype(np.float64) - rp.numpy())
for _ in range(3): m(xc)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10): m(xc)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 10)
m.profile(1)
for _ in range(6): m(xc)
torch.cuda.synchronize()
prof = m.profile_read()
print('%.4f ms | verts max %.3f rms %.4f mm | pose3d max %.3f mm | %s' % (sorted(ts)[2] * 1e3, e.max(), np.sqrt((e ** 2).mean()), ep.max(), {k: round(vv[0] / vv[1], 3) for k, vv in prof.items()}))
'''
for tag, prec, env in [('fp32 build', 'f32', {}),
                       ('16-bit: MDR layers only', 'bf16', {'GATOR_C3_ENCODER': '0', 'GATOR_C3_UPSAMPLE_W1': '0'}),
                       ('16-bit: MDR + encoder', 'bf16', {'GATOR_C3_UPSAMPLE_W1': '0'}),
                       ('16-bit: MDR + regressor weights one plane', 'bf16', {'GATOR_C3_ENCODER': '0'}),
                       ('16-bit: all three (default)', 'bf16', {}),
                       ('fp32 build (again)', 'f32', {})]:
    r = subprocess.run([sys.executable, '-c', CHILD, prec], env=dict(os.environ, **env), capture_output=True, text=True)
    print('%-44s %s' % (tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'ERR ' + r.stderr[-300:]), flush=True)
