#!/bin/bash
# rocprofv3 kernel stats of two builds of the library, one gpurun call:  bash tools/dbg/ab_prof.sh <old .so> <config> <steps>
OLD=$1; CFG=${2:-cartpole_mm}; STEPS=${3:-20}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in $OLD ""; do
  if [ -n "$lib" ]; then export PMBRL_LIB_PATH=$R/$lib; tag=old; else unset PMBRL_LIB_PATH; tag=new; fi
  rm -rf /tmp/kt_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-f32-twin --no-sclk --no-other-configs --repeats 3 > /dev/null 2>&1
  f=$(find /tmp/kt_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $tag"; head -14 $f | cut -d, -f1-4 | cut -c1-150
done
