run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --config $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['value']), round(d['ms_per_step'],4), d.get('kernel_ms'))"; }
for c in cartpole_nomm cartpole_mm dcartpole_mm; do
run base $c
PMBRL_DW_WIDE_MIN=128 run wide128 $c
done
