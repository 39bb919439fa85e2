#!/bin/bash
# MFMA / VALU co-execution counters of the instruction-class microbenchmark (and optionally the bench): gpurun -- bash tools/pmc_coexec.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/pmc_coexec
rm -rf $OUT; mkdir -p $OUT
./tools/microbench/coexec_classes.bin > $OUT/classes_cycles.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/classes -- ./tools/microbench/coexec_classes.bin > $OUT/classes.log 2>&1
cat $OUT/classes_cycles.txt
