#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (the part settles 1-3 % apart between calls):
#   gpurun -- 'bash tools/dbg/ab.sh tools/dbg/old/libpmbrl_r05.so cartpole_mm 20'
OLD=$1; CFG=${2:-cartpole_mm}; STEPS=${3:-20}
for rep in 1 2 3; do
  for lib in $OLD ""; do
    if [ -n "$lib" ]; then export PMBRL_LIB_PATH=$PWD/$lib; tag=old; else unset PMBRL_LIB_PATH; tag=new; fi
    python bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-f32-twin --no-sclk --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$tag', '$CFG', round(d['value']), round(d['ms_per_step'],4), d['kernel_ms'])"
  done
done
