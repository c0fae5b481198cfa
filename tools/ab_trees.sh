#!/bin/bash
# Diagnostic: bench.py of this tree against another checked-out tree (e.g. a git worktree of an older commit with its own built
# library) on ONE box, alternating.  usage: tools/ab_trees.sh <other_tree> [repeats]
set -e
other=$1; n=${2:-2}
mkdir -p gpurun_out
out=$(pwd)/gpurun_out/ab_trees.txt
: > $out
here=$(pwd)
for i in $(seq 1 $n); do
  for t in "$here" "$other"; do
    echo "== $t" >> $out
    (cd $t && timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])") >> $out
  done
done
cat $out
