#!/bin/bash
# Stage-1 legs of bench.py under the Stage1Engine switches, interleaved on one box:  tools/s1_ab.sh [rounds]
#   TNR_S1_CHAIN_WGRAD=0|1 (one chained weight-gradient problem per shared weight)  TNR_S1_TWO_STREAMS=0|1 (body pass on a side stream)
mkdir -p gpurun_out
for r in $(seq 1 ${1:-2}); do
  for cfg in "1 0" "0 0" "1 1"; do
    set -- $cfg
    for leg in "configs[4] stage 1" "stage 1 notebook shape"; do
      TNR_S1_CHAIN_WGRAD=$1 TNR_S1_TWO_STREAMS=$2 python bench.py --leg "$leg" --steps 40 --warmup 10 2>/dev/null |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain=$1 streams=$2  %-24s %8.1f pairs/s  %.3f ms' % (d['leg'], d['value'], d['ms_per_step']))" || exit 1
    done
  done
done
