# A/B of the weight-gradient kernel's ring configuration (one process per setting: the switch is read once), isolated kernels.
# usage (GPU box): bash tools/sweep_tn.sh        -> gpurun_out/sweep_tn/*.log
out=gpurun_out/sweep_tn
mkdir -p $out
for ring in "2,2" "2,3" "1,3" "1,4"; do
  RPE_TN_RING=$ring python tools/bench_conv.py 256 bf16 wgrad > $out/ring_${ring/,/_}.log 2>&1 || exit 1
  tail -1 $out/ring_${ring/,/_}.log
done
RPE_TN_RING=1,4 RPE_TN_WGS=256 python tools/bench_conv.py 256 bf16 wgrad > $out/ring_1_4_wgs256.log 2>&1; tail -1 $out/ring_1_4_wgs256.log
RPE_TN_RING=1,4 RPE_TN_WGS=1024 python tools/bench_conv.py 256 bf16 wgrad > $out/ring_1_4_wgs1024.log 2>&1; tail -1 $out/ring_1_4_wgs1024.log
RPE_TN_RING=2,3 RPE_TN_WGS=256 python tools/bench_conv.py 256 bf16 wgrad > $out/ring_2_3_wgs256.log 2>&1; tail -1 $out/ring_2_3_wgs256.log
