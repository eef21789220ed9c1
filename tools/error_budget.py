"""Where does the fp16 logit error of the HIP path come from?  (CPU tool; restates the student's forward in torch fp32 with a
rounding hook at every tensor the engine stores or feeds to an MFMA in 16 bits.)

For the benchmark's model (4-layer student, hash weights, synthetic MIND-shaped impressions) the logits are computed
  * in fp32 (reference),
  * with ONE site rounded to fp16 at a time (per-source contribution),
  * with all sites rounded (what the engine does), and with selected sites kept in fp32 (what a fix would buy).
Error measure = the tests' own: |logit - ref| / max(1, |ref|), max and r.m.s. over the B x 5 logits.
Sites: W (16-bit weight copies), x0 (embedding output), qkv, P (softmax probabilities as the P.V operand), ctx, h1pre
(attention-output + residual, LayerNorm input), h1, g (GELU output), ypre (FFN output + residual), y (layer output)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import hashinit, synth
from schema import FULL, state_shapes
from oracle import newsrec_oracle as O

torch.set_num_threads(int(os.environ.get("THREADS", os.cpu_count() or 8)))
B, NL, T_, NNEWS = int(os.environ.get("B", 16)), 4, 4, 4000
U, C, L, D, A, H = 50, 5, 30, 256, 12, 768
P = {k: torch.from_numpy(v) for k, v in hashinit.init_state_dict(1234, state_shapes(FULL, NL, D, T_)).items()}
comb = synth.news_table(1234, NNEWS, L).astype(np.int64)
hidx, mask, cidx, label = synth.impressions(1235, B, NNEWS, U, C)
tok = torch.from_numpy(np.concatenate([comb[hidx].reshape(B * U, 2 * L), comb[cidx].reshape(B * C, 2 * L)], 0))
hmask = torch.from_numpy(mask)
DT = torch.float16 if os.environ.get("DTYPE", "fp16") == "fp16" else torch.bfloat16
SITES = ["W", "x0", "qkv", "P", "ctx", "h1pre", "h1", "g", "ypre", "y"]


def forward(round_sites):
    r = lambda name, t: t.to(DT).float() if name in round_sites else t
    # "W": every 16-bit weight copy ; "W:<key>": that one matrix only ; "Wfp32:<key>": every copy EXCEPT that one
    w = lambda k: P[k].to(DT).float() if (("W" in round_sites and "Wfp32:" + k not in round_sites) or "W:" + k in round_sites) else P[k]
    ids, m = tok[:, :L], tok[:, L:].float()
    N = ids.shape[0]
    e = P[O.BERT + "embeddings.word_embeddings.weight"][ids] + P[O.BERT + "embeddings.position_embeddings.weight"][:L][None] \
        + P[O.BERT + "embeddings.token_type_embeddings.weight"][0][None, None]
    ln = lambda x, pfx: torch.nn.functional.layer_norm(x, (H,), P[pfx + ".weight"], P[pfx + ".bias"], 1e-12)
    x = r("x0", ln(e, O.BERT + "embeddings.LayerNorm"))
    madd = ((1.0 - m) * -10000.0)[:, None, None, :]
    rel = torch.from_numpy(O.relpos_bias_table(P[O.BERT + "rel_pos_bias.weight"].numpy(), L))[None]
    for l in range(NL):
        p = O.BERT + "encoder.layer.%d." % l
        lin = lambda t, k: t @ w(p + k + ".weight").T + P[p + k + ".bias"]
        q, k_, v = [r("qkv", lin(x, "attention.self." + n)).reshape(N, L, A, 64).transpose(1, 2) for n in ("query", "key", "value")]
        pr = r("P", torch.softmax(q @ k_.transpose(2, 3) / 8.0 + madd + rel, -1))
        ctx = r("ctx", (pr @ v).transpose(1, 2).reshape(N, L, H))
        h1 = r("h1", ln(r("h1pre", lin(ctx, "attention.output.dense") + x), p + "attention.output.LayerNorm"))
        g = r("g", torch.nn.functional.gelu(lin(h1, "intermediate.dense")))
        x = r("y", ln(r("ypre", lin(g, "output.dense") + h1), p + "output.LayerNorm"))
    # pooling / user encoder / scorer in fp32 as in the engine (the pooling GEMM reads the 16-bit y and the 16-bit fc1 weight copy)
    a = torch.tanh(x @ w(O.PFX + "attn.att_fc1.weight").T + P[O.PFX + "attn.att_fc1.bias"]) @ P[O.PFX + "attn.att_fc2.weight"].T + P[O.PFX + "attn.att_fc2.bias"]
    al = torch.exp(a[..., 0])
    al = al / (al.sum(1, keepdim=True) + 1e-8)
    nv = (al[..., None] * x).sum(1) @ P[O.PFX + "dense.weight"].T + P[O.PFX + "dense.bias"]
    hist, cand = nv[:B * U].reshape(B, U, D), nv[B * U:].reshape(B, C, D)
    ue = "student.user_encoder."
    hv = hist * hmask[..., None] + P[ue + "pad_doc"][None] * (1 - hmask[..., None])
    au = torch.exp((torch.tanh(hv @ P[ue + "attn.att_fc1.weight"].T + P[ue + "attn.att_fc1.bias"]) @ P[ue + "attn.att_fc2.weight"].T
                    + P[ue + "attn.att_fc2.bias"])[..., 0])
    au = au / (au.sum(1, keepdim=True) + 1e-8)
    user = (au[..., None] * hv).sum(1)
    return torch.einsum("bcd,bd->bc", cand, user)


def err(s, ref):
    e = (s - ref).abs() / ref.abs().clamp(min=1.0)
    return float(e.max()), float((e.double() ** 2).mean().sqrt())


t0 = time.time()
with torch.no_grad():
    ref = forward(set())
    print("B = %d (%d logits, |logit| max %.2f), %s activations; reference forward %.0f s" % (B, ref.numel(), float(ref.abs().max()), DT, time.time() - t0), flush=True)
    rows = [("one site: " + s, {s}) for s in SITES]
    rows += [("ALL sites (the engine)", set(SITES)), ("all but h1pre, ypre (fp32 LayerNorm inputs)", set(SITES) - {"h1pre", "ypre"}),
             ("all but y of every layer (fp32 layer outputs)", set(SITES) - {"y"}),
             ("all but h1pre, ypre, h1, y (fp32 residual stream)", set(SITES) - {"h1pre", "ypre", "h1", "y"}),
             ("all but W (fp32 weights)", set(SITES) - {"W"}),
             ("only the GEMM operands' own rounding: W, qkv, P, ctx, g", {"W", "qkv", "P", "ctx", "g"})]
    ss = 0.0
    for name, sites in rows:
        mx, rms = err(forward(sites), ref)
        if name.startswith("one site"):
            ss += rms * rms
        print("%-58s max %.2e  rms %.2e" % (name, mx, rms), flush=True)
    print("root-sum-square of the single-site r.m.s. errors: %.2e" % ss ** 0.5)
    if os.environ.get("PER_GEMM"):
        # which weight matrix carries the weight-rounding error?  One 16-bit copy at a time (everything else fp32), then what a
        # compensated weight (W = W_hi + W_lo, two MFMAs: that GEMM launch at twice the MFMA time) on the top contributors would buy
        keys = []
        for l in range(NL):
            p = O.BERT + "encoder.layer.%d." % l
            keys += [p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight",
                     p + "attention.output.dense.weight", p + "intermediate.dense.weight", p + "output.dense.weight"]
        keys.append(O.PFX + "attn.att_fc1.weight")
        per = []
        for k in keys:
            mx, rms = err(forward({"W:" + k}), ref)
            per.append((rms, mx, k))
            print("only W %-62s max %.2e  rms %.2e" % (k[len(O.BERT):] if k.startswith(O.BERT) else k, mx, rms), flush=True)
        print("root-sum-square over the matrices: %.2e" % sum(r_ * r_ for r_, _, _ in per) ** 0.5)
        per.sort(reverse=True)
        for n in (1, 2, 4, 8):
            keep = {"Wfp32:" + k for _, _, k in per[:n]}
            mx, rms = err(forward(set(SITES) | keep), ref)
            print("the engine with the top %d matrices compensated (exact weights): max %.2e  rms %.2e   [%s]" % (
                n, mx, rms, ", ".join(k.split("encoder.")[-1] for _, _, k in per[:n])), flush=True)
