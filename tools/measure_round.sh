# Round-end measurement set (run on the GPU box through gpurun): PMC HBM traffic, matrix-core busy cycles, rocprofv3 kernel stats + one
# step's timeline, the bench line, the other model families, the fp16 line, the data-parallel path on one GPU.
# usage: bash tools/measure_round.sh <tag> [phases]     outputs under gpurun_out/measure_<tag>/
#   phases (default all, in this order): pmc mfma stats bench models dist loss
# The counter passes run FIRST: in round 4 `rocprofv3 --pmc` aborted (HSA_STATUS_ERROR_INVALID_PACKET_FORMAT, then a hang) twice when it ran
# behind the un-profiled bench and the kernel-trace pass in one call, and never when it ran first on a box (gpurun_out/measure_r04/fetch.err).
tag=${1:-r04}
phases=${2:-"pmc mfma stats bench models dist"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/measure_$tag
mkdir -p $out
# a heartbeat file: the counter passes print nothing for minutes, and a gpurun call that writes nothing for 7 minutes is taken to be hung
( while sleep 45; do date >> $out/heartbeat.txt; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
has() { case " $phases " in *" $1 "*) return 0;; *) return 1;; esac; }
B="--steps 2 --warmup 1 --no-cpu-baseline --precondition-min 2"
cd /tmp && export TMPDIR=/tmp
if has pmc; then
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc -o fetch -- python3 $root/bench.py $B > /dev/null 2> $out/fetch.err || echo "fetch pass failed" >> $out/progress.txt
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc -o write -- python3 $root/bench.py $B > /dev/null 2> $out/write.err || echo "write pass failed" >> $out/progress.txt
  (cd $root && python tools/pmc_reduce.py $out/pmc/fetch_counter_collection.csv $out/pmc/write_counter_collection.csv > $out/hbm_traffic_pmc.json) && echo "pmc done" >> $out/progress.txt && pmc_ok=1
  # the bench phase below looks the dominant kernel's traffic up in profiles/hbm_traffic_pmc.json and withholds it unless that file was measured on
  # the library it loaded (build id): hand it this box's fresh measurement (the caller copies the same file into profiles/ afterwards)
  [ -s $out/hbm_traffic_pmc.json ] && cp $out/hbm_traffic_pmc.json $root/profiles/hbm_traffic_pmc.json
  [ -n "$pmc_ok" ] && rm -rf $out/pmc   # (a pass that aborted -- round 4's second session lost its FETCH pass to HSA_STATUS_ERROR_INVALID_PACKET_FORMAT 5 s in -- leaves the other pass's csv behind)
fi
if has mfma; then
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o m -- python3 $root/bench.py $B > /dev/null 2> $out/mfma.err || echo "mfma pass failed" >> $out/progress.txt
  (cd $root && python tools/mfma_reduce.py $out/mfma/m_counter_collection.csv > $out/mfma_busy_pmc.json) && echo "mfma done" >> $out/progress.txt
  rm -rf $out/mfma
fi
if has stats; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o st -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --precondition-min 2 > $out/bench_under_rocprof.json 2> $out/stats.err
  (cd $root && python tools/timeline.py $out/stats/st_kernel_trace.csv -5 > $out/timeline.txt)
  rm -f $out/stats/*.db $out/stats/st_kernel_trace.csv   # (the raw trace is tens of MB; gpurun merges at most 64 MiB back)
  echo "stats done" >> $out/progress.txt
fi
cd $root
if has bench; then
  python bench.py > $out/bench.json 2> $out/bench.err
  echo "bench done" >> $out/progress.txt
fi
if has models; then
  for m in n td tdo_v2; do python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --precondition-min 2 2>> $out/models.err | tail -1 > $out/bench_model_$m.json; done
  python bench.py --model tdo --depth-head --steps 20 --warmup 5 --no-cpu-baseline --precondition-min 2 2>> $out/models.err | tail -1 > $out/bench_model_tdo_depth.json
  python bench.py --dtype f16 --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>> $out/models.err | tail -1 > $out/bench_f16.json
  echo "models done" >> $out/progress.txt
fi
if has dist; then
  python bench.py --force-dist --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>> $out/dist.err | tail -1 > $out/bench_force_dist.json
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --precondition-min 2 2>> $out/dist.err | tail -1 > $out/bench_plain_same_box.json
  echo "dist done" >> $out/progress.txt
fi
if has loss; then   # (minutes of host time: the convergence records are usually run on their own, tools/loss_record.py 256 300 0 fresh)
  python tools/loss_record.py 32 200 20 > $out/loss_trace.json 2> $out/loss.err
  echo "loss done" >> $out/progress.txt
fi
ls -la $out | tail -30
