#!/bin/bash
# GPU box: rocprofv3 --kernel-trace of the headline and of every other BASELINE configuration (bench.py --leg), each cut to its timed
# steps by tools/trace_window.py -> profiles/<tag>_<name>_kernel_stats.csv ; usage: tools/profile_legs.sh r05 [legs...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r05}; shift
out=gpurun_out/legs_raw; mkdir -p $out profiles
W=3; K=12
run() {   # name, csv suffix, bench arguments...
  name=$1; sfx=$2; shift 2
  rm -rf $out/$sfx
  rocprofv3 --kernel-trace --output-format csv -d $out/$sfx -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline --dedup off --no-other-dtype --no-larger-batch --no-configs --no-kernel-timing "$@" > $out/$sfx.log 2>&1
  echo "== $name"; tail -1 $out/$sfx.log | cut -c1-400
  python3 tools/trace_window.py $out/$sfx $W $K profiles/${tag}_${sfx}_kernel_stats.csv
}
legs=("$@"); [ ${#legs[@]} = 0 ] && legs=(headline configs1 configs4 stage1 stage1nb)
for l in "${legs[@]}"; do
  case $l in
    headline) run "headline" bench ;;
    configs1) run "configs[1]" configs1 --leg "configs[1]" ;;
    configs2) run "configs[2]" configs2 --leg "configs[2]" ;;
    configs4) run "configs[4]" configs4 --leg "configs[4]" ;;
    stage1)   run "configs[4] stage 1" stage1_30_128 --leg "configs[4] stage 1" ;;
    stage1nb) run "stage 1 notebook shape" stage1_24_512 --leg "stage 1 notebook shape" ;;
  esac
done
