for J in 19; do for B in 256 1024 1536 2048; do for T in 1 0; do
GATOR_GAT_TILED=$T python bench.py --steps 6 --warmup 2 --blocks 3 --batch $B --joints $J --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('J=$J B=$B tiled=$T', d['value'], d['roofline']['stages_ms']['gat'])"
done; done; done
