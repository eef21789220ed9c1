"""What the parity tests of the BASELINE configurations MEASURE, beside what they allow; GPU box.
    python tools/parity_measured.py r06      -> profiles/r06_parity_measured.json (+ the raw pytest -s output beside it)
bench.py quotes this file in the `parity` field of each `configs` leg (bounds are the tests' own), and - round 6 - in the headline's
`parity.b32` / `parity.trajectory` (the B = 32 forward and the 50-step trajectory, so that the B = 32 throughput sits beside the
B = 32 error) and in the `quality` object (tests/test_quality_gpu.py): tests print `PARITY_JSON {...}` lines, collected here."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
CASES = {
    "headline": ("tests/test_engine_gpu.py::test_training_step_matches_reference_and_oracle[full_model_1.npz-fp16]", 1e-3, 1e-3),
    "configs[1]": ("tests/test_engine_gpu.py::test_plmnr_finetune_steps[plmnr_full_1.npz-bf16]", 2.8e-2, 1.6e-2),
    "configs[2]": ("tests/test_engine_gpu.py::test_training_step_matches_reference_and_oracle[full_model_2.npz-fp16]", 1e-3, 1e-3),
    "configs[4]": ("tests/test_engine_gpu.py::test_training_step_matches_reference_and_oracle[full_model_5.npz-fp16]", 1e-3, 1e-3),
    "configs[4] stage 1": ("tests/test_stage1_gpu.py::test_stage1_step_matches_notebook_and_oracle[stage1_cfg4.npz-fp16]", 1e-3, 1e-3),
    "stage 1 notebook shape": ("tests/test_stage1_gpu.py::test_stage1_step_matches_notebook_and_oracle[stage1_full.npz-fp16]", 1e-3, 1e-3),
}
out, raw = {}, []
for leg, (node, lb, sb) in CASES.items():
    r = subprocess.run([sys.executable, "-m", "pytest", "-s", "-q", node], capture_output=True, text=True, cwd=ROOT, timeout=900)
    raw.append("==== %s  (%s)\n%s" % (leg, node, r.stdout[-4000:]))
    sc = [(float(a), float(b)) for a, b in re.findall(r"score max\|err\| ([0-9.e+-]+) \(\|ref\| max ([0-9.]+)\)", r.stdout)]
    le = [float(x) for x in re.findall(r"(?:ref [0-9.-]+ err|loss [0-9.-]+ ref [0-9.-]+) ?([0-9.e+-]+)?", r.stdout) if x]
    lo = re.findall(r"loss ([0-9.-]+) ref ([0-9.-]+)", r.stdout)
    le += [abs(float(a) - float(b)) for a, b in lo]
    gr = [float(x) for x in re.findall(r"worst gradient relative L2 error ([0-9.e+-]+)", r.stdout)]
    out[leg] = {"test": node, "passed": r.returncode == 0,
                "logit_bound_rel_to_max1_ref": lb, "loss_bound_rel_to_max1_ref": sb,
                "logit_err_measured_rel_to_max1_ref": max((e / max(1.0, m) for e, m in sc), default=None),
                "loss_err_measured_abs": max(le, default=None), "worst_gradient_rel_l2": max(gr, default=None)}
    print(leg, out[leg], flush=True)
EXTRA = {"b32": "tests/test_bench_shapes_gpu.py::test_b32_forward_matches_oracle",
         "trajectory": "tests/test_engine_gpu.py::test_fifty_training_steps_follow_the_reference_trajectory",
         "quality": "tests/test_quality_gpu.py"}
for name, node in EXTRA.items():
    r = subprocess.run([sys.executable, "-m", "pytest", "-s", "-q", node], capture_output=True, text=True, cwd=ROOT, timeout=900)
    raw.append("==== %s  (%s)\n%s" % (name, node, r.stdout[-6000:]))
    recs = [json.loads(x) for x in re.findall(r"^PARITY_JSON (\{.*\})$", r.stdout, flags=re.M)]
    for rec in recs:
        rec["passed"] = r.returncode == 0
        if name == "quality":
            out.setdefault("quality", {})[rec.pop("key") + "_" + rec["dtype"]] = rec
        elif rec["dtype"] == "fp16":                       # the headline's dtype
            rec.pop("key")
            out["headline"][name] = rec
        else:
            rec.pop("key")
            out["headline"].setdefault("other_dtype", {})[name] = rec
    print(name, [x.get("dtype") for x in recs], "passed" if r.returncode == 0 else "FAILED", flush=True)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_parity_measured.json" % tag), "w"), indent=1)
open(os.path.join(ROOT, "profiles", "%s_parity_measured_pytest.txt" % tag), "w").write("\n".join(raw))
