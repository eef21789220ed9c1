# four / eight waves per 128 x 128 tile (option fine_nw) in the stage-1 step, with and without the second stream
for r in 1 2; do
for ts in 1 0; do
for nw in 4 8; do
 for leg in "configs[4] stage 1" "stage 1 notebook shape"; do
  TNR_S1_TWO_STREAMS=$ts python bench.py --leg "$leg" --steps 40 --warmup 10 --gemm-opt fine_nw=$nw 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$ts fine_nw=$nw  %-24s %8.1f pairs/s  %.3f ms' % (d['leg'], d['value'], d['ms_per_step']))"
 done
done
done
done
