# Round-end measurement set (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, PMC HBM traffic.
# usage: bash tools/measure_round.sh <tag>     outputs under gpurun_out/measure_<tag>/
set -e
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/measure_$tag
mkdir -p $out
cd $root
python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc -o fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc -o write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/write.err
cd $root
python tools/pmc_reduce.py $out/pmc/fetch_counter_collection.csv $out/pmc/write_counter_collection.csv 6 > $out/hbm_traffic_pmc.json
ls -la $out $out/stats | tail -20
