#!/bin/bash
# rocprofv3 kernel stats of the configurations other than the headline one (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_kernel_stats.sh r03e'
# -> gpurun_out/<tag>/kstats_<config>.csv + the bench line of the profiled run
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cartpole_mm dcartpole_mm stress32 stress32_mm; do
  steps=10; [ "$c" = stress32 ] && steps=3; [ "$c" = stress32_mm ] && steps=3
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- \
    python $R/bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-f32-twin > $O/kstats_bench_$c.json 2>$O/kstats_$c.err
  f=$(find $O/kt_$c -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/kstats_$c.csv
  rm -rf $O/kt_$c
done
ls -la $O | grep kstats
