# the optimiser update on its own stream under the next forward's frozen layers (Engine.overlap_update), interleaved on one box
for r in 1 2 3; do
for ov in 0 1; do
  TNR_OVERLAP_UPDATE=$ov python bench.py --steps 100 --warmup 20 --no-cpu-baseline --dedup off --no-other-dtype --no-larger-batch --no-configs --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$ov  %8.1f impressions/s  %.4f ms  loss %s' % (d['value'], d['ms_per_step'], d.get('final_loss')))"
done
done
