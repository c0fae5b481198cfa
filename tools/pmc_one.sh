# usage: bash tools/pmc_one.sh <tag> <one_conv args...>   -- SQ counters for ONE conv shape (run on the GPU box)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/one_conv.py "$@" > $out/log.txt 2>&1
