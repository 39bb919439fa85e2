#!/bin/bash
# Training-step throughput over batch sizes on one box (hipGraph replay; tools/train_bench.py):  bash tools/train_sweep.sh > profiles/<tag>_train_batch_sweep.txt
for j in 17 19; do
for b in 16 32 64 128 256 512 1024; do
  python tools/train_bench.py --no-cpu-baseline --no-eager --batch $b --joints $j --steps 8 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('J=%d B=%4d  %7.3f ms/step  %8.1f samples/s  %.2f TFLOP/s algorithmic' % (d['joints'], d['batch'], d['graph']['ms_per_step'], d['value'], d['roofline']['achieved']))"
done; done
