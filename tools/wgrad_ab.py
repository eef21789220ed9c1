"""Interleaved same-process timing of tnr_gemm_tn_wgrad over the number of M-splits (development aid; GPU box)."""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tiny-newsrec_amd"))
import torch, tnr_hip as T
dev = "cuda:0"
M = int(os.environ.get("M", 52800))
Mp = (M + 63) // 64 * 64
for (N, K) in ((3072, 768), (768, 3072), (2304, 768), (768, 768), (256, 768)):
    dy = torch.zeros((Mp, N), device=dev, dtype=torch.bfloat16); dy[:M] = torch.randn((M, N), device=dev).to(torch.bfloat16)
    x = torch.zeros((Mp, K), device=dev, dtype=torch.bfloat16); x[:M] = torch.randn((M, K), device=dev).to(torch.bfloat16)
    dw = torch.zeros((N, K), device=dev)
    tiles = (N // 256) * (K // 256) if (N % 256 == 0 and K % 256 == 0) else (N // 128) * (K // 128)
    one_round = max(1, min(64, 256 // tiles))
    cands = sorted({max(1, one_round // 2), one_round, min(64, one_round * 2), max(1, (one_round * 3) // 4)})
    ws = torch.zeros(T.query("tnr_gemm_tn_ws_elems", N, K, max(cands)), device=dev)
    acc = collections.defaultdict(list)
    for rnd in range(6):
        for sp in (cands if rnd % 2 == 0 else cands[::-1]):
            run = lambda: T.call("tnr_gemm_tn_wgrad", dy, N, x, K, dw, K, M, N, K, ws, sp, 0)
            run(); run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            acc[sp].append(e0.elapsed_time(e1) * 100)
    print("dW %4d x %4d (tiles %2d, one round = %2d splits): " % (N, K, tiles, one_round) +
          "  ".join("%d: %.1f us" % (sp, sorted(acc[sp])[3]) for sp in cands))
