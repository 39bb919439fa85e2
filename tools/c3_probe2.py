"""Config-3 speed probe at B=2048 J=19: variants by environment (read at ctx creation) in one process."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model

def timed(fn, steps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    return sorted(ts)[2]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.from_numpy(synthetic.synthetic_pose2d(B, 19, seed=31)).cuda()
ref = None
for tag, env, prec, sub in [
        ('f32', {}, 'f32', 0),
        ('c3 2 waves', {'GATOR_C3_WAVES': '2'}, 'bf16', 0),
        ('c3 3 waves', {'GATOR_C3_WAVES': '3'}, 'bf16', 0),
        ('c3 2 waves chunk 512', {'GATOR_C3_WAVES': '2', 'GATOR_MDR_PERSIST_CHUNK': '512'}, 'bf16', 0),
        ('c3 3 waves chunk 384', {'GATOR_C3_WAVES': '3', 'GATOR_MDR_PERSIST_CHUNK': '384'}, 'bf16', 0),
        ('c3 2 waves, 2 sub-batch streams', {'GATOR_C3_WAVES': '2'}, 'bf16', 2),
        ('c3 3 waves, 2 sub-batch streams', {'GATOR_C3_WAVES': '3'}, 'bf16', 2),
        ('f32, 2 sub-batch streams', {}, 'f32', 2),
        ('c3 2 waves (again)', {'GATOR_C3_WAVES': '2'}, 'bf16', 0)]:
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    z, m = build_model('coco19_alpha', 'fused')
    m.precision = prec
    if sub: m.subbatch_streams = sub
    v, p = m(x)
    torch.cuda.synchronize()
    if tag == 'c3 2 waves': ref = v.clone()
    same = '' if ref is None or prec != 'bf16' else (' | bits == c3 2 waves' if torch.equal(v, ref) else ' | max diff vs c3 2 waves %.3e mm' % (float((v - ref).abs().max()) * 1e3))
    dt = timed(lambda: m(x))
    print('%-36s %9.1f meshes/s  %.4f ms%s' % (tag, B / dt, dt * 1e3, same), flush=True)
    del m
    for k, vv in old.items():
        if vv is None: os.environ.pop(k, None)
        else: os.environ[k] = vv
