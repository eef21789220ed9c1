"""Summarise rocprofv3 --pmc csv output: per kernel (short id) and grid, mean counter value per dispatch."""
import csv, glob, re, sys, collections
root = sys.argv[1]
pats = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"([A-Za-z_0-9]+_kernel)", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"][:30]) + " grid=" + r["Grid_Size"]
        if pats and not any(p in name for p in pats):
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    print(name)
    print("   " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
