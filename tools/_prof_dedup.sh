cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_dedup; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --dedup only --steps 20 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1
tail -1 $out/trace.log | cut -c1-300
python3 - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/prof_dedup/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms/step %.3f"%(tot/25/1e6))
for r in rows[:28]:
    m=re.search(r"([A-Za-z_0-9]+_kernel)",r["Name"]); n=m.group(1) if m else r["Name"][:40]
    print("%-34s calls/step %5.1f avg %8.2f us  %5.2f%%"%(n,int(r["Calls"])/25,float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY
