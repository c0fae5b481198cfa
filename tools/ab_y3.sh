#!/bin/bash
# Diagnostic: A/B of the y3-free bottleneck dataflow on one box, alternating: default (y3-free, layers 1-2) / RPE_Y3_KEEP=1 (new forward,
# y3 still written, round-3 backward) / RPE_NO_Y3FREE=1 (round-3 dataflow).  usage: tools/ab_y3.sh [repeats] [extra env assignments...]
n=${1:-2}
mkdir -p gpurun_out
out=gpurun_out/ab_y3.txt
: > $out
run() {
  echo "== $1" >> $out
  env $1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('final_loss'))" >> $out
}
for i in $(seq 1 $n); do
  run "RPE_AB=default"
  run "RPE_Y3_KEEP=1"
  run "RPE_NO_Y3FREE=1"
done
cat $out
