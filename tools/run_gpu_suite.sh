# The whole GPU test suite in one process with a heartbeat file (a gpurun call that writes nothing for 7 minutes is taken to be hung): bash tools/run_gpu_suite.sh [pytest args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/full
mkdir -p $out
( while sleep 60; do date >> $out/heartbeat.txt; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
cd $root
timeout -k 10 1100 python -m pytest tests -q -m gpu "$@" > $out/gpu_tests.log 2>&1
rc=$?
tail -5 $out/gpu_tests.log
exit $rc
