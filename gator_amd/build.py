"""Build libgator_hip.so (gfx950) in-tree with hipcc.  `python -m gator_amd.build` or gator_amd.build.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libgator_hip.so')
SOURCES = ['api.hip', 'basic_kernels.hip', 'fused_api.hip', 'fused_pack.hip', 'upsample_fused.hip', 'mdr_fused.hip', 'gat_fused.hip', 'gat_roles.hip', 'gat_tiled.hip', 'gat_tail.hip', 'upsample_bf16.hip', 'upsample_x3.hip', 'upsample_x2.hip', 'caller_kernels.hip', 'train_ops.hip',
           'graph_consts.cpp', 'comm_rccl.cpp']
HEADERS = ['internal.h', os.path.join(ROOT, 'include', 'gator_hip.h'), os.path.join(ROOT, 'include', 'gator_train.h')]
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
# Measured on gfx950 (tools/microbench/coissue.hip): up to six plain VALU instructions (v_fma_f32, v_cvt_pk_bf16_f32, v_cndmask
# ...) issue in the shadow of one bf16 MFMA for ~0.6 cycles each, while a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 never hides
# and adds 6-8 cycles.  Building the MFMA kernels without the packed forms (the flags below) was tried and is NOT used: the
# VALU-bound parts (exact GELU, operand splits) lose more than the MFMA-adjacent parts gain (1.00 -> 1.03 ms per step).
NO_PK = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
NO_PK_SOURCES = tuple(x for x in os.environ.get('GATOR_NO_PK', '').split(',') if x)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


DIAG_LIB = os.path.join(HERE, 'lib', 'libgator_hip_diag.so')


def build(force=False, verbose=True, diag=False):
    """Production library: no diagnostics compiled in, and a hot kernel that spills registers is a build error
    (gator_amd/kernel_resources.py).  diag=True builds libgator_hip_diag.so instead (-DGATOR_DIAG: in-kernel cycle stamps read
    through GATOR_GAT_STAMPS / GATOR_MDR_STAMPS; select it with GATOR_AMD_LIB=<path>, see tools/gat_stamps.py)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(os.path.join(HERE, 'lib'), exist_ok=True)
    objdir = os.path.join(HERE, 'lib', 'obj_diag' if diag else 'obj')
    lib = DIAG_LIB if diag else LIB
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    hdrs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.hpp', '.cuh'))]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        is_hip = s.endswith('.hip')
        extra = os.environ.get('GATOR_HIPCC_FLAGS_' + s.rsplit('.', 1)[0], '').split()      # A/B builds of one source (e.g. -fno-slp-vectorize)
        # the per-source extra flags are part of an object's identity: a stamp beside it records them, and a change (set or unset)
        # forces a rebuild -- an A/B object is never silently reused by a production build, nor the other way round
        stamp = obj + '.flags'
        want = ' '.join(extra + (NO_PK if s in NO_PK_SOURCES else []))
        have = open(stamp).read() if os.path.exists(stamp) else ''
        if force or want != have or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + extra + (['-DGATOR_DIAG=1'] if diag else []) + (NO_PK if s in NO_PK_SOURCES else []) + \
                (['-x', 'hip', '-Rpass-analysis=kernel-resource-usage'] if is_hip else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            for stale in (obj, stamp):       # a failed compile must not leave the other flag set's object behind a matching stamp
                if os.path.exists(stale):
                    os.remove(stale)
            procs.append((s, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True), stamp, want))
    from . import kernel_resources as kr
    remarks = {}
    failed = None
    for s, obj, p, stamp, want in procs:
        err = p.communicate()[1]
        if p.returncode != 0:
            sys.stderr.write(err)
            failed = failed or s
            continue
        with open(stamp, 'w') as fh:       # the stamp records the flags of an object that exists
            fh.write(want)
        noise = ('remark:', '-Rpass-analysis')
        rest = [ln for ln in err.splitlines() if ln.strip() and not any(n in ln for n in noise)]
        rest = [ln for ln in rest if not (ln.lstrip().startswith('|') or ln.lstrip()[:1].isdigit() and ' | ' in ln)]
        if rest and verbose:
            sys.stderr.write('\n'.join(rest) + '\n')
        remarks[s] = (obj, kr.parse(err))
    if failed:
        raise RuntimeError('hipcc failed on %s' % failed)
    if not diag:
        try:
            kr.check({s: r for s, (_, r) in remarks.items()})
        except RuntimeError:
            for s, (obj, _) in remarks.items():      # do not leave objects of a rejected build behind
                if os.path.exists(obj):
                    os.remove(obj)
            raise
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', lib] + objs + ['-ldl']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, diag='--diag' in sys.argv))
