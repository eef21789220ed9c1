# the four-stage 128 x 128 loop (option fine_st) in the stage-1 step, with and without the second stream
for r in 1 2; do
for ts in 0 1; do
for st in 2 4; do
 for leg in "configs[4] stage 1"; do
  TNR_S1_TWO_STREAMS=$ts python bench.py --leg "$leg" --steps 40 --warmup 10 --gemm-opt fine_st=$st 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$ts fine_st=$st  %-24s %8.1f pairs/s  %.3f ms' % (d['leg'], d['value'], d['ms_per_step']))"
 done
done
done
done
