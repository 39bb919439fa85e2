#!/bin/bash
# Effective clock of the MDR launch and the vertex regressor in the round-6 form of the path and in the round-5 form (two tail launches, whole head), one box:
# GRBM_GUI_ACTIVE per dispatch (PMC pass) over the kernel's average duration (kernel-trace pass of the same command).  DESIGN section 5.
#   bash tools/clock_by_form.sh   (through gpurun; output gpurun_out/clock_by_form/)
OUT=$PWD/gpurun_out/clock_by_form; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
ARGS="bench.py --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 20 --warmup 5 --blocks 5"
for form in r06 r05; do
  if [ $form = r05 ]; then export GATOR_GAT8_TAIL=0 GATOR_MDR_HEAD_PARTIALS=0; else unset GATOR_GAT8_TAIL GATOR_MDR_HEAD_PARTIALS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$form" -- python3 $ARGS > "$OUT/trace_$form.log" 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$form" -- python3 $ARGS > "$OUT/pmc_$form.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, statistics
out = sys.argv[1]
def short(n):
    m = re.search(r'(k_[a-z0-9_]+)', n); return m.group(1) if m else n[:30]
for form in ('r06', 'r05'):
    st = {}
    for r in csv.DictReader(open(glob.glob(out + '/trace_%s/**/*kernel_stats.csv' % form, recursive=True)[0])):
        st[short(r['Name'])] = float(r['AverageNs']) / 1e3
    cyc = {}
    for r in csv.DictReader(open(glob.glob(out + '/pmc_%s/**/*counter_collection.csv' % form, recursive=True)[0])):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            cyc.setdefault(short(r['Kernel_Name']), []).append(float(r['Counter_Value']))
    for k in ('k_gat8', 'k_gat_lifter', 'k_gat_joint', 'k_mdr_persist', 'k_mdr_head', 'k_mdr_head_finish', 'k_upsample_x2'):
        if k in st and k in cyc:
            c = statistics.median(cyc[k]) / 8.0          # the counter sums the eight XCDs
            print('%s  %-20s %8.1f us (trace)  %10.0f cycles per XCD (pmc pass)  -> %.3f GHz if the pmc pass ran at the trace pass speed' % (form, k, st[k], c, c / st[k] / 1e3))
PY
