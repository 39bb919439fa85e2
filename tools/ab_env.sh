#!/bin/bash
# A/B of library switches read at gator_create on ONE box: bench.py (no CPU leg, no variants) per setting, interleaved twice.
# usage: tools/ab_env.sh out_dir "TAG1:VAR=VAL VAR2=VAL" "TAG2:" ...      (an empty setting = the default library)
OUT=$1; shift
mkdir -p "$OUT"
for rep in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 --blocks 7 --min-timed-s 1 ${AB_ARGS:-} > "$OUT/$tag.$rep.json" 2> "$OUT/$tag.$rep.err"
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
