#!/bin/bash
# Collect one round's measurement set on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r01h'
# Everything lands in gpurun_out/<tag>/ ; copy what is to be kept into profiles/ (see profiles/README.md).
# Counters are collected in their own passes (never together with a trace), every command has its own timeout.
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (default flags: with the CPU baseline leg), its hardware counters measured by the same command
#    (bench.py --pmc: nested rocprofv3 --pmc passes, one per counter set; writes profiles/pmc_<config>_<precision>.json
#    with the library's build id)
timeout 1500 python $R/bench.py --pmc > $O/bench_cartpole_nomm.json 2>$O/bench.err
# 2. kernel trace + stats of the same command (shorter run, no CPU leg)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- \
  python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-twin --no-other-configs > $O/bench_under_rocprof.json 2>$O/kt.err
# 3. the other configurations, each with its counters
for c in cartpole_mm dcartpole_mm stress32 stress32_mm; do
  timeout 900 python $R/bench.py --pmc --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-f32-twin > $O/bench_$c.json 2>$O/bench_$c.err
done
cp $R/profiles/pmc_*.json $O/
# 4. in-kernel cycle stamps, the other configurations, the neighbouring paths
timeout 120 python $R/tools/phase_prof.py > $O/phase_prof.txt 2>/dev/null
for c in cartpole_mm dcartpole_mm stress32 stress32_mm; do timeout 120 python $R/tools/phase_prof.py $c > $O/phase_prof_$c.txt 2>/dev/null; done
timeout 300 python $R/tools/bench_bnn.py > $O/bench_bnn.json 2>/dev/null
timeout 300 python $R/tools/mcp_speed.py > $O/mcp_speed.txt 2>/dev/null
timeout 300 python $R/tools/mcp_example_shape.py > $O/mcp_example_shape.txt 2>/dev/null
timeout 300 python $R/tools/bench_single_group.py > $O/single_group.txt 2>/dev/null
timeout 120 python $R/tools/phase_prof.py cartpole_mm 0 100 25 0 > $O/phase_prof_cartpole_mm_g1.txt 2>/dev/null
# 5. kernel stats of the other configurations (one rocprofv3 pass each, no counters)
for c in cartpole_mm dcartpole_mm stress32 stress32_mm; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$c -- \
    python $R/bench.py --config $c --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline --no-f32-twin > /dev/null 2>$O/ks_$c.err
  f=$(find $O/ks_$c -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $O/kstats_$c.csv; fi
  rm -rf $O/ks_$c
done
# flatten the rocprof outputs (they sit under <dir>/<host>/<pid>_*.csv)
for d in kt; do
  for f in $(find $O/$d -name '*.csv' 2>/dev/null); do cp $f $O/${d}_$(basename $f | sed 's/^[0-9]*_//'); done
done
rm -rf $O/kt
ls -la $O
cat $O/bench_cartpole_nomm.json
head -8 $O/kt_kernel_stats.csv | cut -c1-160
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/pmc_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['build_id'])
    for k,v in d['kernels'].items(): print('  %-28s hbm %.3e B  mfma %s%%' % (k, v['hbm_bytes_per_launch'] or 0, v['mfma_util_pct']))
PY
