#!/bin/bash
# Batch sweep of the full forward on one GPU: J=17 fp32 at several batch sizes, config 3 and the eval mode.
for B in 64 128 256 512 1024 2048 4096; do
python bench.py --steps 10 --warmup 3 --blocks 5 --batch $B --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('J=17 f32 B=$B', d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
done
python bench.py --steps 10 --warmup 3 --blocks 5 --batch 2048 --joints 19 --precision bf16 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config3 J=19 bf16 B=2048', d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
python bench.py --steps 10 --warmup 3 --blocks 5 --batch 2048 --joints 19 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('J=19 f32 B=2048', d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
python bench.py --steps 10 --warmup 3 --blocks 5 --batch 1024 --no-cpu-baseline --mode eval 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eval mode J=17 B=1024', d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
python bench.py --steps 10 --warmup 3 --blocks 5 --batch 1024 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gather mode J=17 B=1024', d['value'], d['ms_per_step'])"
GATOR_GAT_X3=0 GATOR_MDR_X3=0 GATOR_UPSAMPLE_X3=0 python bench.py --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('all fp32-input MFMA B=256', d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
GATOR_MDR_X3=1 python bench.py --steps 10 --warmup 3 --blocks 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('all bf16x3 (MDR_X3=1) B=256', d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"
