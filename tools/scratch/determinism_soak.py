"""Two identical runs of the bench workload (N steps each): every loss and the final parameters must agree bit for bit -
the tile queue changes WHICH workgroup computes a tile, never the result.   python tools/scratch/determinism_soak.py [steps]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch
import engine as E, hashinit, synth
from schema import FULL, state_shapes
dev, B, N_NEWS, seed = "cuda:0", 32, 51282, 1234
S = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
tables = torch.from_numpy(synth.teacher_tables(seed, 4, N_NEWS, cfg.D)).to(dev)
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, 40 * B, N_NEWS, cfg.U, cfg.C)]
def run():
    eng = E.Engine(cfg, dev, max_batch=B, dtype="fp16")
    eng.load_state_dict(hashinit.init_state_dict(seed, state_shapes(FULL, 4, cfg.D, 4)))
    losses = []
    for i in range(S):
        s = slice((i % 40) * B, (i % 40 + 1) * B)
        l, _ = eng.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables)
        losses.append(l.clone())
        eng.backward()
        eng.step(lr=1e-4)
    torch.cuda.synchronize()
    h = hashlib.sha256(eng.flat[True].cpu().numpy().tobytes()).hexdigest()[:16]
    return torch.stack(losses).cpu(), h
l0, h0 = run()
l1, h1 = run()
print("steps %d: losses bit-identical %s ; parameter hash %s vs %s ; last total loss %.5f" % (S, bool(torch.equal(l0, l1)), h0, h1, float(l0[-1][0])))
assert torch.equal(l0, l1) and h0 == h1
