run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --config $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['value']), round(d['ms_per_step'],4), d.get('kernel_ms'), d['config']['rows_per_wg'], d['config']['workgroups'])"; }
PMBRL_MM_PARTS=1 run whole cartpole_mm
run default cartpole_mm
PMBRL_MM_PARTS=1 run whole dcartpole_mm
run default dcartpole_mm
PMBRL_MM_PARTS=4 run parts4 dcartpole_mm
