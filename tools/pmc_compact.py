"""Compact a rocprofv3 counter_collection.csv (one row per dispatch and counter: MBs) to per-kernel averages of the forward's kernels.
usage: python tools/pmc_compact.py in.csv out.csv"""
import csv, re, sys, collections


def short(name):
    m = re.search(r'(k_[a-z0-9_]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:40]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        if 'gator' not in r['Kernel_Name']:
            continue
        k = short(r['Kernel_Name'])
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    w = csv.writer(open(sys.argv[2], 'w'))
    w.writerow(['kernel', 'dispatches', 'avg_duration_us', 'counter', 'avg_value_per_dispatch'])
    for k, cs in acc.items():
        if len(next(iter(cs.values()))) < 4:      # set-up kernels of gator_create
            continue
        for c, v in sorted(cs.items()):
            w.writerow([k, len(v), round(sum(dur[k]) / len(dur[k]) / 1e3, 2), c, round(sum(v) / len(v))])


if __name__ == '__main__':
    main()
