"""gpurun_out/profiles_raw -> profiles/r01_*.{csv,json,md}: per-kernel time table of the bench command and the
HBM traffic / MFMA activity of the dominant kernel from the PMC passes (gfx950 corrections applied)."""
import collections, csv, glob, json, os, re, sys
raw, out = "gpurun_out/profiles_raw", "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
sys.path.insert(0, "tiny-newsrec_amd")
import tnr_hip
src_sha16 = tnr_hip.source_sha16()         # ties the summary to the library SOURCES (a rebuild changes the binary hash, not this)
def short(n):
    m = re.search(r"([A-Za-z_0-9]+_kernel)", n)
    return m.group(1) if m else n.split("(")[0][-50:]
def shortt(n):                 # with the template arguments (tile height, epilogue flag set)
    m = re.search(r"([A-Za-z_0-9]+_kernel(<[^>]*>)?)", n)
    return m.group(1) if m else n.split("(")[0][-50:]
stats = glob.glob(raw + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
steps = 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# (the per-kernel table of the timed steps is written by tools/trace_window.py since round 5: <tag>_bench_kernel_stats.csv, cut to the
# timed window; this script keeps the PMC summaries)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(raw + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dom = "gemm_nt_pp_kernel"
c = {k: sum(v) / len(v) for k, v in agg[dom].items()}
dom_rows = [r for r in rows if short(r["Name"]) == dom]      # the <MI=8> and <MI=7> instantiations of the one kernel
dom_calls = sum(int(r["Calls"]) for r in dom_rows)
dom_ns = sum(float(r["TotalDurationNs"]) for r in dom_rows)
res = {"src_sha16": src_sha16, "dtype": dtype, "kernel": dom + " (all template instantiations; = every tnr_gemm_nt launch bench.py times)",
       "avg_launch_us_trace": dom_ns / dom_calls / 1e3, "launches_in_trace": dom_calls,
       "counters_mean_per_launch": c,
       # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of wide
       # coalesced reads (16 B/lane global_load and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-B stores
       "hbm_bytes_per_launch": (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024,
       "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1),
       "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)}
json.dump(res, open("%s/%s_gemm_nt_pmc.json" % (out, tag), "w"), indent=1)
print("step kernel time %.3f ms ; %s avg %.1f us ; HBM bytes/launch %.3e ; L2 hit %.2f ; MFMA busy %.2f" % (
    tot / steps / 1e6, dom, res["avg_launch_us_trace"], res["hbm_bytes_per_launch"], res["l2_hit_rate"], res["mfma_busy_frac"]))

# the other kernels of the step, same counters (mean per launch).  FETCH_SIZE is doubled as for the GEMM where the kernel reads
# with 16 bytes per lane (all of these do since round 3 except the attention kernels' 8-byte fragment-shaped reads, marked).
others = {}
raw_agg = collections.defaultdict(lambda: collections.defaultdict(list))      # by full name: the mangled ones keep their prefix in short()
for f in glob.glob(raw + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        raw_agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("gemm_tn_rs_kernel", "gemm_tn_pp_kernel", "slab_reduce_kernel", "attn_fwd_kernel", "attn_bwd_kernel", "ln_fwd_kernel", "ln_bwd_kernel", "embed_ln_kernel",
          "attpool_fwd_kernel", "attpool_bwd_kernel", "amsgrad_kernel", "sgemm_group_kernel", "user_fwd_fused_kernel"):
    vals = collections.defaultdict(list)
    for name, cs in raw_agg.items():
        if k in name:
            for n, v in cs.items():
                vals[n] += v
    kr = [r for r in rows if k in r["Name"]]
    calls, ns = sum(int(r["Calls"]) for r in kr), sum(float(r["TotalDurationNs"]) for r in kr)
    if not calls or not vals:
        continue
    ck = {n: sum(v) / len(v) for n, v in vals.items()}
    us = ns / calls / 1e3
    byt = (2 * ck.get("FETCH_SIZE", 0) + ck.get("WRITE_SIZE", 0)) * 1024
    others[k] = {"avg_launch_us_trace": round(us, 2), "launches_per_step": round(calls / steps, 1),
                 "fabric_bytes_per_launch": round(byt), "tb_per_s": round(byt / us / 1e6, 2),
                 "l2_hit_rate": round(ck.get("TCC_HIT_sum", 0) / max(ck.get("TCC_HIT_sum", 0) + ck.get("TCC_MISS_sum", 0), 1), 3),
                 "mfma_busy_frac": round(ck.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(ck.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), 3),
                 "lds_bank_conflict_cycles": round(ck.get("SQ_LDS_BANK_CONFLICT", 0)),
                 "fetch_calibrated": k not in ("attn_fwd_kernel", "attn_bwd_kernel")}
json.dump({"src_sha16": src_sha16, "dtype": dtype, "kernels": others}, open("%s/%s_other_kernels_pmc.json" % (out, tag), "w"), indent=1)
for k, v in others.items():
    print("%-24s %8.1f us x %4.1f  %7.1f MB  %5.2f TB/s  L2 hit %.2f  MFMA busy %.2f" % (k, v["avg_launch_us_trace"], v["launches_per_step"],
          v["fabric_bytes_per_launch"] / 1e6, v["tb_per_s"], v["l2_hit_rate"], v["mfma_busy_frac"]))
