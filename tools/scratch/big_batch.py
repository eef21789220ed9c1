"""B = 128 / 256 per GPU (M = 211 200 / 422 400 token rows: byte offsets beyond 2^31, element counts beyond 2^30): the scores of one
big batch == the scores of its B = 32 pieces bit for bit, its gradient == the mean of theirs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import torch, engine as E, hashinit, synth
from schema import FULL, state_shapes
dev, N_NEWS, seed = "cuda:0", 51282, 1234
BB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = E.EngineConfig(n_layers=4, trainable_layers=(2, 3), num_teachers=4)
sd = hashinit.init_state_dict(seed, state_shapes(FULL, 4, cfg.D, 4))
comb = torch.from_numpy(synth.news_table(seed, N_NEWS, cfg.L)).to(dev)
tables = torch.from_numpy(synth.teacher_tables(seed, 4, N_NEWS, cfg.D)).to(dev)
hidx, mask, cidx, label = [torch.from_numpy(x).to(dev) for x in synth.impressions(seed + 1, BB, N_NEWS, cfg.U, cfg.C)]
small = E.Engine(cfg, dev, max_batch=32, dtype="fp16"); small.load_state_dict(sd)
scores, g = [], torch.zeros_like(small.flat_g)
for i in range(BB // 32):
    s = slice(32 * i, 32 * i + 32)
    _, sc = small.forward_indexed(comb, hidx[s], mask[s], cidx[s], label[s], tables)
    scores.append(sc.clone()); small.backward(); g += small.flat_g
g /= BB // 32
del small; torch.cuda.empty_cache()
big = E.Engine(cfg, dev, max_batch=BB, dtype="fp16"); big.load_state_dict(sd)
_, sc = big.forward_indexed(comb, hidx, mask, cidx, label, tables)
big.backward(); torch.cuda.synchronize()
print("B=%d: scores bit-identical to the B=32 pieces: %s ; gradient vs mean of pieces: rel L2 %.2e ; mem %.1f GB" % (
    BB, bool(torch.equal(sc, torch.cat(scores))), float((big.flat_g - g).norm() / g.norm()), torch.cuda.max_memory_allocated() / 1e9))
