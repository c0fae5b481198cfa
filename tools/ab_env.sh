#!/bin/bash
# Diagnostic: A/B one environment switch over bench.py runs on one box.  usage: tools/ab_env.sh VAR v1 v2 ... ('-' = unset) (results in gpurun_out/ab_VAR.txt)
set -e
var=$1; shift
mkdir -p gpurun_out
out=gpurun_out/ab_$var.txt
: > $out
for v in "$@"; do
  echo "== $var=$v" >> $out
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi   # "-": the variable is unset for this run
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out
done
cat $out
