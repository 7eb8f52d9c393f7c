for e in "PMBRL_REG_BWD=0" "PMBRL_REG=0" "PMBRL_MM_XCH=0"; do
  echo "== $e"
  env $e timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_mm_parts.py -m gpu -q -k "oracle or parts" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-230 | head -24
done
