# HBM traffic per kernel symbol of the train step from two rocprofv3 PMC passes (run on the GPU box): usage: bash tools/pmc_traffic.sh <out.json>
set -e
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$root/gpurun_out/hbm_traffic_pmc.json}
tmp=$root/gpurun_out/pmc_tmp
mkdir -p $tmp $(dirname $out)
( while sleep 45; do date >> $tmp/heartbeat.txt; done ) &    # (a gpurun call that writes nothing for 7 minutes is taken to be hung)
hb=$!
trap "kill $hb 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $tmp -o fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2 > /dev/null 2> $tmp/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $tmp -o write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2 > /dev/null 2> $tmp/write.err
cd $root
python tools/pmc_reduce.py $tmp/fetch_counter_collection.csv $tmp/write_counter_collection.csv > $out
rm -rf $tmp
