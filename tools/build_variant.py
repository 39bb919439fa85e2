"""A/B builds: libgator_hip.so with ONE source recompiled under extra flags, linked against the production objects of the others.
    python tools/build_variant.py <name> <source.hip> [--file other.hip] [flags...]   ->  gator_amd/lib/variants/libgator_<name>.so
(--file: compile that file in place of csrc/<source.hip>, e.g. an older revision saved with `git show HEAD:... > /tmp/x.hip`)
Run a variant with GATOR_AMD_LIB=<that path>; several variants can be measured in one gpurun call (tools/ab_bench.sh)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gator_amd import build as b

def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    path = os.path.join(b.CSRC, src)
    if flags[:1] == ['--file']:
        path, flags = flags[1], flags[2:]
    b.build(verbose=False)
    vdir = os.path.join(b.HERE, 'lib', 'variants')
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, '%s_%s.o' % (name, src.rsplit('.', 1)[0]))
    cmd = ['/opt/rocm/bin/hipcc'] + b.FLAGS + flags + ['-x', 'hip', '-c', path, '-o', obj]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    objs = [os.path.join(b.HERE, 'lib', 'obj', s.rsplit('.', 1)[0] + '.o') for s in b.SOURCES if s != src] + [obj]
    lib = os.path.join(vdir, 'libgator_%s.so' % name)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-shared', '-fPIC', '--offload-arch=gfx950', '-o', lib] + objs + ['-ldl'])
    os.remove(obj)
    print(lib)

if __name__ == '__main__':
    main()
