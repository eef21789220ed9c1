#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the default bench command + separate PMC passes
# (HBM fetch / write bytes; SQ activity) as MI355X_MICROARCH.md prescribes.  Outputs under gpurun_out/profiles_raw/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles_raw; rm -rf $out; mkdir -p $out
if [ "$1" = "trace" ]; then PMC=0; else PMC=1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dedup off --no-other-dtype --no-larger-batch --no-configs > $out/trace.log 2>&1
i=0
[ $PMC = 1 ] && for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --dedup off --no-other-dtype --no-larger-batch --no-configs > $out/pmc$i.log 2>&1
  i=$((i+1))
done
tail -1 $out/trace.log
