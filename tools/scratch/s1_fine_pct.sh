for r in 1 2; do
for fp in 60 40 25 10; do
 for leg in "configs[4] stage 1" "stage 1 notebook shape"; do
  python bench.py --leg "$leg" --steps 40 --warmup 10 --gemm-opt fine_pct=$fp 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fine_pct=$fp  %-24s %8.1f pairs/s  %.3f ms' % (d['leg'], d['value'], d['ms_per_step']))"
 done
done
done
