#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench command, then separate PMC passes (never combined with a trace domain).
# Output: gpurun_out/prof_<tag>/...; digest with tools/pmc_digest.py.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 20 --warmup 5 --blocks 5 > "$OUT/stats.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 5 --warmup 2 --blocks 2 > "$OUT/pmc_$name.log" 2>&1
done
find "$OUT" -name "*.csv" | head -30
# round 5: the same evidence for BASELINE config 3 (B = 2048, J = 19, 16-bit operand mode): kernel-trace stats and the instruction / stall counters
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c3" -- python3 bench.py --config 3 --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 10 --warmup 3 --blocks 3 > "$OUT/stats_c3.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_c3_stalls" -- python3 bench.py --config 3 --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 3 --warmup 2 --blocks 2 > "$OUT/pmc_c3_stalls.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_c3_mfma" -- python3 bench.py --config 3 --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 3 --warmup 2 --blocks 2 > "$OUT/pmc_c3_mfma.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_stalls" -- python3 bench.py --no-cpu-baseline --no-variants --no-config3 --min-timed-s 0 --steps 5 --warmup 2 --blocks 2 > "$OUT/pmc_stalls.log" 2>&1
find "$OUT" -name "*.csv" | wc -l
