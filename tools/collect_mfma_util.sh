#!/bin/bash
# MFMA busy cycles and MFMA op counts from the hardware counters (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_mfma_util.sh r03e'
# One rocprofv3 --pmc pass per counter set and configuration (counters never together with a trace).
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cartpole_nomm cartpole_mm stress32; do
  steps=6; [ "$c" = stress32 ] && steps=2
  for set in "MfmaUtil" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${c}_$n -- \
      python $R/bench.py --config $c --steps $steps --warmup 1 --no-cpu-baseline --no-f32-twin --timing-steps 1 > /dev/null 2>$O/pmc_${c}_$n.err
    f=$(find $O/pmc_${c}_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp $f $O/mfma_${c}_$n.csv
    rm -rf $O/pmc_${c}_$n
  done
done
ls -la $O | grep mfma_
