#!/bin/bash
# Diagnostic: bench.py under several environment settings on ONE box.  usage: tools/ab_multi.sh "base" "VAR=1" "A=1 B=2" ...
set -e
mkdir -p gpurun_out
out=gpurun_out/ab_multi.txt
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  if [ "$cfg" = "base" ]; then pre=""; else pre="env $cfg"; fi
  $pre timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out
done
cat $out
