#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the bench step with a GEMM option at two values (in-step per-kernel times).
#   bash tools/trace_ab.sh pp 0 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
key=$1; shift
for v in "$@"; do
  out=gpurun_out/trace_${key}_$v; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dedup off --no-other-dtype --no-larger-batch --no-kernel-timing --gemm-opt $key=$v > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1].split("/")[1], "kernel time per step %.3f ms" % (tot / 25 / 1e6))
for r in rows[:14]:
    n = r["Name"]; m = re.search(r"([A-Za-z_0-9]+_kernel(<[^>]*>)?)", n)
    print("  %-44s calls/step %5.1f  avg %8.2f us  %5.1f %%" % ((m.group(1) if m else n[:44])[:44], int(r["Calls"]) / 25, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
