# Round-end measurement set (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, PMC HBM traffic, matrix-core busy
# cycles, one step's timeline, the other model families, the data-parallel path on one GPU, the fp16 line, the loss record.
# usage: bash tools/measure_round.sh <tag>     outputs under gpurun_out/measure_<tag>/
set -e
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/measure_$tag
mkdir -p $out
cd $root
python bench.py > $out/bench.json 2> $out/bench.err
echo "bench done" > $out/progress.txt
cd /tmp && export TMPDIR=/tmp
# a heartbeat file: the counter passes print nothing for minutes, and a gpurun call that writes nothing for 7 minutes is taken to be hung
( while sleep 45; do date >> $out/heartbeat.txt; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
# Under rocprofv3's counter collection every kernel is serialised, so the engine's second-stream probe (does a kernel on the candidate
# overlap with one on the caller's stream?) can only fail and would create all four candidates; the first PMC pass of round 4 aborted with
# HSA_STATUS_ERROR_INVALID_PACKET_FORMAT in that configuration (gpurun_out/measure_r04/fetch.err).  The profiled passes take the first stream.
export RPE_NO_SIDE_PROBE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --precondition-min 2 > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done" >> $out/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc -o fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2 > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc -o write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2 > /dev/null 2> $out/write.err
echo "pmc done" >> $out/progress.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o m -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2 > /dev/null 2> $out/mfma.err
echo "mfma done" >> $out/progress.txt
cd $root
unset RPE_NO_SIDE_PROBE
python tools/pmc_reduce.py $out/pmc/fetch_counter_collection.csv $out/pmc/write_counter_collection.csv > $out/hbm_traffic_pmc.json
python tools/mfma_reduce.py $out/mfma/m_counter_collection.csv > $out/mfma_busy_pmc.json
python tools/timeline.py $out/stats/st_kernel_trace.csv -5 > $out/timeline.txt
for m in n td tdo_v2; do python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --precondition-min 2 2>> $out/models.err | tail -1 > $out/bench_$m.json; done
python bench.py --model tdo --depth-head --steps 20 --warmup 5 --no-cpu-baseline --precondition-min 2 2>> $out/models.err | tail -1 > $out/bench_tdo_depth.json
echo "models done" >> $out/progress.txt
python bench.py --dtype f16 --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>> $out/models.err | tail -1 > $out/bench_f16.json
python bench.py --force-dist --steps 30 --warmup 8 --precondition-min 2 2>> $out/dist.err | tail -1 > $out/bench_force_dist.json
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>> $out/dist.err | tail -1 > $out/bench_plain_same_box.json
echo "dist done" >> $out/progress.txt
if [ "${RPE_MEASURE_LOSS:-0}" = "1" ]; then   # (the convergence records take minutes of host time: tools/loss_record.py 256 300 0 fresh, run on their own)
python tools/loss_record.py 32 200 20 > $out/loss_trace.json 2> $out/loss.err
echo "loss done" >> $out/progress.txt
fi
rm -rf $out/pmc $out/mfma $out/stats/*.db 2>/dev/null || true   # (the raw counter tables are tens of MB; gpurun merges at most 64 MiB back)
ls -la $out | tail -30
