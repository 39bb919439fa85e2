"""Build-time check of the hot kernels' register budget: parse hipcc's -Rpass-analysis=kernel-resource-usage remarks and fail
when a kernel named in HOT spills to scratch.  `python -m gator_amd.kernel_resources` prints the table for every kernel."""
import os
import re
import subprocess
import sys

# Kernels of the DEFAULT paths (timed forward at every batch size, evaluation mode): scratch (spilled registers) beyond the
# budget given here is a build error; the budget is 0 unless a kernel is listed in SCRATCH_BUDGET with the bytes per lane it is
# known to carry (so that it cannot silently grow).  The fp32-input-MFMA forms behind the GATOR_*_X3=0 switches (k_gat<false>,
# k_mdr_layer<*, 0>, k_mdr_persist<0>) and the all-bf16x3 form (k_mdr_layer<*, 1>, k_mdr_persist<1>) are A/B variants, not checked.
HOT = ('k_gat<true', 'k_gat8', 'k_gat_lifter', 'k_gat_joint', 'k_gat_tiled<', 'k_mdr_layer<0, 2>', 'k_mdr_layer<1, 2>', 'k_mdr_layer<2, 2>', 'k_mdr_persist<2>',
       'k_mdr_layer<0, 3>', 'k_mdr_layer<1, 3>', 'k_mdr_layer<2, 3>', 'k_mdr_persist<3>',
       'k_mdr_head<', 'k_mdr_head_finish', 'k_upsample_x3', 'k_upsample_x2', 'k_upsample_bf16', 'k_regress', 'k_jreg_reduce', 'k_joint_errors', 'k_rigid_align', 'k_preprocess')
SCRATCH_BUDGET = {}      # (round 6: k_gat_tiled's entries are gone -- its scratch was a hoisted sum, gat_tiled.hip: aggregate)
_FIELD = re.compile(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)')
_NAME = re.compile(r'remark:\s+Function Name: (\S+)')


def demangle(names):
    if not names:
        return {}
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([tool] + list(names), capture_output=True, text=True).stdout.split('\n')
            if len(out) >= len(names):
                return dict(zip(names, out))
        except OSError:
            continue
    return {n: n for n in names}


def parse(stderr):
    """-> {mangled name: {'VGPRs': int, 'AGPRs': int, 'TotalSGPRs': int, 'ScratchSize': int, 'Occupancy': int, 'LDS Size': int}}"""
    res, cur = {}, None
    for ln in stderr.splitlines():
        m = _NAME.search(ln)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = _FIELD.search(ln)
        if m and cur is not None:
            try:
                cur[m.group(1).strip()] = int(m.group(2))
            except ValueError:
                pass
    return res


def short_name(demangled):
    s = re.sub(r'^void ', '', demangled)
    s = s.replace('gator::(anonymous namespace)::', '').replace('gator::', '')
    return re.sub(r'\(.*$', '', s)


def check(remarks_by_source):
    """remarks_by_source: {source: parsed remarks}.  Raises if a HOT kernel has scratch; returns the rows."""
    rows, bad = [], []
    for source, res in remarks_by_source.items():
        names = demangle(list(res))
        for mangled, f in res.items():
            short = short_name(names.get(mangled, mangled))
            rows.append((source, short, f.get('VGPRs', 0), f.get('AGPRs', 0), f.get('TotalSGPRs', 0),
                         f.get('ScratchSize', 0), f.get('Occupancy', 0), f.get('LDS Size', 0)))
            hot = any(short.startswith(h) for h in HOT)
            if short.startswith('_Z'):           # c++filt could not demangle it (e.g. __bf16 parameters): match the bare kernel name
                hot = any(re.search(r'\d+%s(?![a-z_])' % re.escape(h.split('<')[0]), short) for h in HOT)
            budget = max([v for k, v in SCRATCH_BUDGET.items() if short.startswith(k)] or [0])
            if f.get('ScratchSize', 0) > budget and hot:
                bad.append('%s: %s spills %d bytes/lane to scratch (budget %d)' % (source, short, f['ScratchSize'], budget))
    if bad:
        raise RuntimeError('register spills in hot kernels:\n  ' + '\n  '.join(bad))
    return rows


def main():
    from . import build as _b
    res = {}
    for s in _b.SOURCES:
        if s.endswith('.hip'):
            src = os.path.join(_b.CSRC, s)
            cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + _b.FLAGS + ['-x', 'hip', '-c', src, '-o', os.devnull,
                                                                               '-Rpass-analysis=kernel-resource-usage']
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed on %s:\n%s' % (s, r.stderr[-2000:]))
            res[s] = parse(r.stderr)
    try:
        rows = check(res)
        err = None
    except RuntimeError as e:
        rows, err = [], e
        for source, rr in res.items():
            names = demangle(list(rr))
            for mangled, f in rr.items():
                rows.append((source, short_name(names.get(mangled, mangled)), f.get('VGPRs', 0), f.get('AGPRs', 0), f.get('TotalSGPRs', 0),
                             f.get('ScratchSize', 0), f.get('Occupancy', 0), f.get('LDS Size', 0)))
    print('%-22s %-44s %5s %5s %5s %8s %4s %8s' % ('source', 'kernel', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'occ', 'LDS'))
    for r in sorted(rows):
        print('%-22s %-44s %5d %5d %5d %8d %4d %8d' % r)
    if err:
        print(err)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
