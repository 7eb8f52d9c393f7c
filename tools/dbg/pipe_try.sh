for v in off 10,30 10,10,20 8,8,8,16 6,6,6,6,16 12,12,16 5,5,5,5,5,15; do
  echo "== PMBRL_DW_PIPE=$v"
  PMBRL_DW_PIPE=$v timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-f32-twin --no-sclk 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
done
