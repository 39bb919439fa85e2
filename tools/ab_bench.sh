#!/bin/bash
# Measure several builds of the library on ONE box: bench.py (no CPU leg, no variants) per library, interleaved twice.
# usage: tools/ab_bench.sh out_dir lib1.so lib2.so ...   (default library: pass "default")
OUT=$1; shift
mkdir -p "$OUT"
for rep in 1 2; do
  for lib in "$@"; do
    tag=$(basename "$lib" .so)
    if [ "$lib" = "default" ]; then unset GATOR_AMD_LIB; else export GATOR_AMD_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 --blocks 7 ${AB_ARGS:-} > "$OUT/$tag.$rep.json" 2> "$OUT/$tag.$rep.err"
    python - "$OUT/$tag.$rep.json" "$tag" "$rep" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-28s rep %s: %9.1f meshes/s  %.4f ms  stages %s' % (sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['roofline'].get('stages_ms')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
  done
done
