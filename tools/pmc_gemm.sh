#!/bin/bash
# Run on the GPU box: PMC passes (separate, with --kernel-trace only) over a few NT GEMM launches.   bash tools/pmc_gemm.sh pp=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_gemm; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 tools/gemm_few.py "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
# launches differ by shape; group by dispatch order within kernel instead: print means per kernel
for key, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(key)
    for k in sorted(m): print("   %-28s %14.0f   %6.3f of WAVE_CYCLES" % (k, m[k], m[k] / wc))
    if "TCC_HIT_sum" in m: print("   L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
PY
