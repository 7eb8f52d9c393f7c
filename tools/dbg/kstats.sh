#!/bin/bash
# rocprofv3 kernel stats of one bench configuration, names shortened:  bash tools/dbg/kstats.sh <config> <steps>
CFG=${1:-stress32}; STEPS=${2:-3}
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt_one
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_one -- python $R/bench.py --config $CFG --steps $STEPS --warmup 2 --repeats 1 --no-cpu-baseline --no-f32-twin --no-sclk --no-other-configs > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/kt_one/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('%-62s calls %5s avg %10.1f us  %5s %%' % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
