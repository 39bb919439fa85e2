for B in 256 512 1024 2048; do for T in 1 0; do
GATOR_GAT_TAIL=$T python bench.py --steps 10 --warmup 3 --blocks 5 --batch $B --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B tail=$T', d['value'], d['roofline']['stages_ms'])"
done; done
