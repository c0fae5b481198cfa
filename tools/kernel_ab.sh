#!/bin/bash
# Diagnostic: per-kernel-symbol time per step (bench.py's HIP-event profile pass) under several environment settings.  usage: tools/kernel_ab.sh "-" "VAR=1" ...
mkdir -p gpurun_out
out=gpurun_out/kernel_ab.txt
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  if [ "$cfg" = "-" ]; then e="RPE_AB=none"; else e="$cfg"; fi
  env $e RPE_BENCH_TOPK=80 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precondition-min 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'])
for k in d['roofline']['kernels']:
    print('%-44s x%-3d %.3f ms  %s GB/s %s TF/s' % (k['kernel'], k['launches_per_step'], k['ms_per_step'], k['gbs'], k['tflops']))
" >> $out
done
cat $out
