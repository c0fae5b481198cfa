# One train step's launches in start order from a rocprofv3 kernel trace (run on the GPU box): bash tools/trace_list.sh [out_dir] [bench args...]
set -e
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$root/gpurun_out/trace_list}
case $out in /*) ;; *) out=$root/$out ;; esac
shift || true
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -o st -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --precondition-min 2 "$@" > $out/bench.json 2> $out/bench.err
cd $root
python tools/timeline.py $out/tr/st_kernel_trace.csv -5 --list > $out/list.txt
rm -rf $out/tr
