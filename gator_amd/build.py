"""Build libgator_hip.so (gfx950) in-tree with hipcc.  `python -m gator_amd.build` or gator_amd.build.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libgator_hip.so')
SOURCES = ['api.hip', 'basic_kernels.hip', 'fused_api.hip', 'fused_pack.hip', 'upsample_fused.hip', 'mdr_fused.hip', 'gat_fused.hip', 'upsample_bf16.hip', 'upsample_x3.hip',
           'graph_consts.cpp']
HEADERS = ['internal.h', os.path.join(ROOT, 'include', 'gator_hip.h')]
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(os.path.join(HERE, 'lib'), exist_ok=True)
    objdir = os.path.join(HERE, 'lib', 'obj')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    hdrs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.hpp', '.cuh'))]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + (['-x', 'hip'] if s.endswith('.hip') else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % s)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
