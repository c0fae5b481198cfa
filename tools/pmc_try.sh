#!/bin/bash
# Diagnostic: one rocprofv3 counter pass over a short bench run, with a heartbeat file (the pass prints nothing for minutes).
# usage: bash tools/pmc_try.sh <counter> [env assignments...]    -> gpurun_out/pmc_try/
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_try
mkdir -p $out
ctr=$1; shift
( while sleep 45; do date >> $out/heartbeat.txt; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 700 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc -o pass -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2 > /dev/null 2> $out/pass.err
echo "rc $?"
tail -3 $out/pass.err
ls -la $out/pmc | head -5
wc -l $out/pmc/pass_counter_collection.csv 2>/dev/null
rm -rf $out/pmc
