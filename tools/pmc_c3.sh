#!/bin/bash
# PMC passes of the B=256 J=17 forward in fp32 and in 16-bit mode (config 3 arithmetic), per kernel: gpurun -- bash tools/pmc_c3.sh
set -u
OUT=$PWD/gpurun_out/pmc_c3
rm -rf "$OUT"; mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
for prec in f32 bf16; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/${prec}_$i" -- python3 bench.py --precision $prec --no-cpu-baseline --no-variants --steps 5 --warmup 2 --blocks 2 > "$OUT/${prec}_$i.log" 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('OUT', 'gpurun_out/pmc_c3')
for prec in ('f32', 'bf16'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/pmc_c3/%s_*/**/*counter_collection.csv' % prec, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void gator::(anonymous namespace)::', '')
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', prec)
    for k, cs in acc.items():
        if 'mdr_persist' in k or 'gat8' in k:
            print(k[:40], {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
