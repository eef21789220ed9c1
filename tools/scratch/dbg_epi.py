import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch, tnr_hip as T
T.lib().tnr_gemm_set_option(b"allow_fine", 0)
M, N, K = 256, 256, 64
A = np.zeros((M, K), np.float32); A[:, 0] = 1.0
B = np.zeros((N, K), np.float32); B[:, 0] = np.arange(N)          # C[m][n] = n
a, b = torch.from_numpy(A).cuda().bfloat16(), torch.from_numpy(B).cuda().bfloat16()
c = torch.full((M, N), -1.0, device="cuda")
T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, T.EPI_OUTF32)
torch.cuda.synchronize()
print("route", T.query("tnr_gemm_nt_route", M, N, K, T.EPI_OUTF32))
print("row 0 :", c[0, :70].cpu().numpy().astype(int))
A[:, 0] = np.arange(M); B[:, 0] = 1.0                             # C[m][n] = m
a, b = torch.from_numpy(A).cuda().bfloat16(), torch.from_numpy(B).cuda().bfloat16()
T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, T.EPI_OUTF32)
torch.cuda.synchronize()
print("col 0 :", c[:40, 0].cpu().numpy().astype(int), c[120:136, 0].cpu().numpy().astype(int))
print("col 17:", c[:20, 17].cpu().numpy().astype(int))
