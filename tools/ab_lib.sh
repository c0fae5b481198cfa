#!/bin/bash
# Diagnostic: bench.py with this tree's library against ANOTHER BUILD of the same ABI (e.g. the previous commit's, built in a git
# worktree and copied to ab/librpe_prev.so) on ONE box, alternating.  usage: tools/ab_lib.sh <other.so> [repeats] [bench args...]
set -e
other=$(readlink -f $1); n=${2:-3}; shift; shift || true
mkdir -p gpurun_out
out=gpurun_out/ab_lib.txt
: > $out
for i in $(seq 1 $n); do
  for lib in "" "$other"; do
    echo "== ${lib:-this tree}" >> $out
    RPE_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out
  done
done
cat $out
