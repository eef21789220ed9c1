"""Per-kernel PMC summary of one bench leg from tools/pmc.sh passes (+ a kernel trace of the same command for durations):
    tools/pmc.sh gpurun_out/pmc_s1 python3 bench.py --leg "stage 1 notebook shape" --steps 4 --warmup 2 --no-cpu-baseline
    python tools/pmc_leg.py gpurun_out/pmc_s1 profiles/r05_stage1_24_512_pmc.json attn_long attpool
-> mean per launch of every kernel whose name contains one of the given substrings: duration (from the passes' own kernel
traces), fabric bytes (FETCH_SIZE in KiB doubled for 16-byte-per-lane reads + WRITE_SIZE in KiB: MI355X_MICROARCH.md, HBM section),
TB/s, L2 hit rate, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections, csv, glob, json, re, sys
root, out = sys.argv[1], sys.argv[2]
pats = sys.argv[3:]
short = lambda n: (re.search(r"([A-Za-z_0-9]+_kernel)", n) or [None, n[:40]])[1] if re.search(r"([A-Za-z_0-9]+_kernel)", n) else n[:40]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(p in k for p in pats):
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(p in k for p in pats):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, cs in sorted(cnt.items()):
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    us = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else None
    byt = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
    res[k] = {"median_launch_us_under_pmc": us, "launches_seen": len(dur[k]), "fabric_bytes_per_launch": round(byt),
              "tb_per_s": round(byt / us / 1e6, 2) if us else None,
              "l2_hit_rate": round(c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1), 3),
              "mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), 3),
              "lds_bank_conflict_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0)), "counters_mean_per_launch": c}
    print("%-34s %8.1f us  %8.1f MB  %5s TB/s  L2 hit %.2f  MFMA busy %.2f" % (k, us or 0, byt / 1e6, res[k]["tb_per_s"], res[k]["l2_hit_rate"], res[k]["mfma_busy_frac"]))
json.dump(res, open(out, "w"), indent=1)
