#!/bin/bash
# Diagnostic: alternate bench.py over several environment settings on one box.  usage: tools/ab_envs.sh <repeats> "VAR=1" "OTHER=1 X=2" ...   ("-" = nothing set)
n=$1; shift
mkdir -p gpurun_out
out=gpurun_out/ab_envs.txt
: > $out
for i in $(seq 1 $n); do
  for cfg in "$@"; do
    echo "== $cfg" >> $out
    if [ "$cfg" = "-" ]; then e="RPE_AB=none"; else e="$cfg"; fi
    env $e timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['host_issue_ms_per_step'])" >> $out
  done
done
cat $out
