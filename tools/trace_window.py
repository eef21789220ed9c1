"""A rocprofv3 --kernel-trace of `bench.py` (headline or --leg ...) cut to its TIMED steps; GPU box or here.
    python tools/trace_window.py <dir with *kernel_trace.csv> <warmup> <steps> <out.csv> [anchor substring = embed_ln]
The anchor kernel (the embedding + LayerNorm kernel opens every encoder pass) divides the dispatch list into steps: n anchors per
step = anchors / (warmup + steps) must be whole.  Step i = [start of its first anchor, start of step i + 1's first anchor): whatever
the process launched in between - library kernels, ATen fills, copies - belongs to it.  Steps warmup .. warmup + steps - 2 are used
(the last timed step has no successor to close its window).  Written: per kernel calls per step (exact), average duration, share;
printed / returned: launches per step, kernel time per step (sum of durations), how much of it ran beside another kernel (stage 1
runs its two passes on two streams), wall per step and the idle time (wall minus the union of the kernels' intervals)."""
import collections, csv, glob, json, re, sys


def short(n):
    if "at::native" in n or "at::cuda" in n:             # torch's own kernels keep enough of their name to be recognised as such
        m = re.search(r"(at::native::[A-Za-z_0-9]+(<[^,>]*)?)", n)
        f = re.search(r"at::native::([A-Za-z_0-9]*Functor[A-Za-z_0-9_]*|direct_copy[A-Za-z_0-9_]*|[a-z_]+_cuda_out)", n)
        return "ATen " + (m.group(1) if m else "kernel") + (" / " + f.group(1) if f else "")
    m = re.search(r"([A-Za-z_0-9]+_kernel(<[^>]*>)?)", n)
    if m:
        return m.group(1)
    m = re.search(r"(at::native::[A-Za-z_0-9:]+|copyBuffer[A-Za-z_0-9]*|fillBuffer[A-Za-z_0-9]*|Cijk_[A-Za-z0-9_]{0,40})", n)
    return m.group(1) if m else n.split("(")[0][-60:]


def window(trace_dir, warmup, steps, out_csv=None, anchor="embed_ln"):
    f = glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if anchor in r[2]]
    per = len(anchors) / float(warmup + steps)
    assert per >= 1 and abs(per - round(per)) < 1e-9, "anchor %r: %d dispatches over %d steps" % (anchor, len(anchors), warmup + steps)
    per = int(round(per))
    first = [anchors[i * per] for i in range(warmup + steps)]
    used = list(range(warmup, warmup + steps - 1))
    agg = collections.OrderedDict()
    wall = ktime = launches = busy = 0.0
    for i in used:
        a, b = first[i], first[i + 1]
        wall += rows[b][0] - rows[a][0]
        cur_s = cur_e = None                     # union of the kernels' intervals: with two streams (stage 1) kernels run side by side
        for s_, e_, n in rows[a:b]:
            if cur_e is None or s_ > cur_e:
                busy += (cur_e - cur_s) if cur_e is not None else 0
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += (cur_e - cur_s) if cur_e is not None else 0
        for s_, e_, n in rows[a:b]:
            k = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
            k[0] += 1; k[1] += e_ - s_; k[2] = min(k[2], e_ - s_); k[3] = max(k[3], e_ - s_)
            ktime += e_ - s_
            launches += 1
    n = float(len(used))
    res = {"steps_used": len(used), "launches_per_step": launches / n, "kernel_ms_per_step": ktime / n / 1e6, "wall_ms_per_step": wall / n / 1e6,
           "busy_ms_per_step": busy / n / 1e6, "idle_ms_per_step": (wall - busy) / n / 1e6, "side_by_side_ms_per_step": (ktime - busy) / n / 1e6, "aten_or_copy_rows": [k for k in agg if k.startswith("ATen ") or (not k.endswith("_kernel") and "_kernel<" not in k)]}
    if out_csv:
        with open(out_csv, "w") as fo:
            w = csv.writer(fo)
            w.writerow(["kernel", "calls_per_step", "us_per_step", "avg_us", "min_us", "max_us", "percent_of_kernel_time"])
            for k, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, "%.2f" % (c / n), "%.1f" % (t / n / 1e3), "%.2f" % (t / c / 1e3), "%.2f" % (mn / 1e3), "%.2f" % (mx / 1e3), "%.2f" % (100.0 * t / ktime)])
            w.writerow(["# timed steps %d (of %d), launches/step %.1f, kernel ms/step %.4f (sum of durations; %.4f of it beside another kernel), "
                        "wall ms/step %.4f, idle ms/step %.4f" % (len(used), steps, res["launches_per_step"], res["kernel_ms_per_step"],
                                                                  res["side_by_side_ms_per_step"], res["wall_ms_per_step"], res["idle_ms_per_step"])])
    return res, agg


if __name__ == "__main__":
    res, agg = window(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None, sys.argv[5] if len(sys.argv) > 5 else "embed_ln")
    print(json.dumps(res))
    for k, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-52s %6.2f / step  %9.1f us / step  avg %8.2f us" % (k[:52], c / res["steps_used"], t / res["steps_used"] / 1e3, t / c / 1e3))
