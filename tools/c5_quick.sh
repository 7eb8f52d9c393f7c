#!/bin/bash
# quick C5 check on the GPU box: bench line (3 steps) + in-kernel phase profile -> gpurun_out/<tag>/
TAG=${1:-rXX}; SUF=${2:-a}
O=gpurun_out/$TAG; mkdir -p $O
python bench.py --config stress32 --steps 3 --warmup 2 --no-cpu-baseline --no-f32-twin > $O/bench_stress32_$SUF.json 2> $O/bench_stress32_$SUF.err
python -c "
import json;d=json.load(open('$O/bench_stress32_$SUF.json'));print(d['value'],d['ms_per_step'],d['kernel_ms'])"
python tools/phase_prof.py stress32 > $O/phase_prof_stress32_$SUF.txt 2>&1; tail -45 $O/phase_prof_stress32_$SUF.txt
