"""Digest of tools/profile_round.sh's output: per-kernel average duration (kernel trace), HBM-side bytes per launch
(FETCH_SIZE x 2 as the gfx950 guide prescribes for wide coalesced reads, + WRITE_SIZE; both in KiB units -> bytes), MFMA
instruction counts and MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).
Writes profiles/<tag>_* (kernel stats CSV, one CSV per PMC pass, the summary JSON)."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src = os.path.join('gpurun_out', 'prof_' + tag)
KEEP = ('k_gat', 'k_mdr_layer', 'k_mdr_persist', 'k_mdr_head', 'k_upsample', 'k_mdr_joint', 'k_pack_vc', 'k_jreg')


def short(name):
    m = re.match(r'_ZN5gator\d+_GLOBAL__N_1\d+(k_[a-z0-9_]+?)[EI]', name)      # (I: a template argument list follows the name)      # a signature rocprofv3 could not demangle (_Float16 arguments)
    if m:
        return m.group(1)
    n = name.replace('(anonymous namespace)::', '').replace('void ', '').replace('gator::', '')
    return n.split('(')[0].strip()


def find(pattern):
    r = glob.glob(os.path.join(src, pattern), recursive=True)      # gpurun MERGES into gpurun_out/: keep the newest run
    return max(r, key=os.path.getmtime) if r else None


def pmc_rows(path):
    """-> {kernel: {counter: [values per dispatch]}}"""
    out = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        k = short(row['Kernel_Name'])
        if not k.startswith(KEEP):
            continue
        out[k][row['Counter_Name']].append(float(row['Counter_Value']))
    return out


def main():
    os.makedirs('profiles', exist_ok=True)
    stats = find('stats/**/*kernel_stats.csv')
    trace = find('stats/**/*kernel_trace.csv')
    summary = {}
    if stats:
        shutil.copy(stats, 'profiles/%s_kernel_stats_bench_B256.csv' % tag)
    if trace:
        dur = defaultdict(list)
        for row in csv.DictReader(open(trace)):
            k = short(row['Kernel_Name'])
            if k.startswith(KEEP):
                dur[k].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
        for k, v in dur.items():
            v = v[len(v) // 4:]          # drop the warm-up quarter
            summary[k] = {'kernel': k, 'launches': len(v), 'avg_us': round(sum(v) / len(v), 2)}
    for name in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU_MFMA_BF16', 'SQ_INSTS_VALU'):
        path = find('pmc_%s/**/*counter_collection.csv' % name)
        if not path:
            continue
        shutil.copy(path, 'profiles/%s_pmc_%s.csv' % (tag, name if name in ('FETCH_SIZE', 'WRITE_SIZE') else ('MFMA' if 'MFMA' in name else 'VALU')))
        for k, ctr in pmc_rows(path).items():
            d = summary.setdefault(k, {'kernel': k})
            avg = {c: sum(v) / len(v) for c, v in ctr.items()}
            if 'FETCH_SIZE' in avg:
                d['fetch_MB_raw'] = round(avg['FETCH_SIZE'] / 1024, 3)
                d['fetch_MB_corrected'] = round(2 * avg['FETCH_SIZE'] / 1024, 3)
            if 'WRITE_SIZE' in avg:
                d['write_MB'] = round(avg['WRITE_SIZE'] / 1024, 3)
            if 'SQ_INSTS_VALU_MFMA_BF16' in avg:
                d['mfma_bf16_insts'] = int(avg['SQ_INSTS_VALU_MFMA_BF16'])
                d['mfma_f16_insts'] = int(avg.get('SQ_INSTS_VALU_MFMA_F16', 0))
                d['mfma_f32_insts'] = int(avg.get('SQ_INSTS_VALU_MFMA_F32', 0))
                if avg.get('GRBM_GUI_ACTIVE'):
                    d['mfma_busy_frac'] = round(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (avg['GRBM_GUI_ACTIVE'] / 8), 4)
            if 'SQ_INSTS_VALU' in avg:
                d['valu_insts'] = int(avg['SQ_INSTS_VALU'])
                if avg.get('SQ_BUSY_CYCLES'):
                    d['valu_active_over_busy'] = round(avg.get('SQ_ACTIVE_INST_VALU', 0) / avg['SQ_BUSY_CYCLES'], 4)
    rows = sorted(summary.values(), key=lambda r: -r.get('avg_us', 0))
    json.dump(rows, open('profiles/%s_pmc_summary_B256.json' % tag, 'w'), indent=1)
    # where the digest comes from: the commit checked out when it was made (the library that ran is built from it) -- bench.py
    # prints it beside every digest-derived field
    try:
        import subprocess
        head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(['git', 'status', '--porcelain', '--', 'gator_amd/csrc'], capture_output=True, text=True).stdout.strip())
    except OSError:
        head, dirty = None, None
    json.dump({'digest': '%s_pmc_summary_B256.json' % tag, 'commit': head, 'kernel_sources_modified': dirty,
               'command': 'tools/profile_round.sh %s (bench.py --no-cpu-baseline --no-variants at B = 256)' % tag},
              open('profiles/%s_pmc_provenance.json' % tag, 'w'), indent=1)
    for r in rows:
        print(r)


if __name__ == '__main__':
    main()
