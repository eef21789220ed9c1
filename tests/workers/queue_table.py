"""Worker (own process: it fills the library's (device, stream) table for good): the persistent GEMM on more streams than the
table has slots.  Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tiny-newsrec_amd"))
import numpy as np, torch, tnr_hip as T

dev, M, N, K = "cuda:0", 3000, 512, 128          # 12 x 2 tiles of 256 x 256 ...
T.lib().tnr_gemm_set_option(b"allow_fine", 0)    # ... on the persistent (queue-fed) kernel although they fill few CUs
assert T.query("tnr_gemm_nt_route", M, N, K, T.EPI_OUTF32) in (T.ROUTE_256, T.ROUTE_224), "shape must take the queue-fed kernel"
rs = np.random.RandomState(0)
A, B = rs.randint(-3, 4, (M, K)).astype(np.float32), rs.randint(-3, 4, (N, K)).astype(np.float32)
a, b = torch.from_numpy(A).to(dev).to(torch.bfloat16), torch.from_numpy(B).to(dev).to(torch.bfloat16)
want = torch.from_numpy(A @ B.T).to(dev)
L = T.lib()


def gemm_ok(stream):
    c = torch.zeros((M, N), device=dev)
    with torch.cuda.stream(stream):
        T.call("tnr_gemm_nt", a, K, b, K, c, N, M, N, K, None, None, 0, None, 0, T.EPI_OUTF32)
    stream.synchronize()
    return bool(torch.equal(c, want))


# torch.cuda.Stream() hands out 32 pooled streams per device: distinct ones come from the runtime itself
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]


def new_stream():
    h = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(h)) == 0
    return torch.cuda.ExternalStream(h.value)


gemm_ok(torch.cuda.current_stream())             # the default stream takes a slot like any other
streams, ok, refused_at, msg = [], 1, None, ""
for i in range(1, 140):
    s = new_stream()
    streams.append(s)
    try:
        assert gemm_ok(s)
        ok += 1
    except T.TnrError as e:
        refused_at, msg = i, str(e)
        break
# the streams bound so far keep working, twice each (counters back at zero after every launch), and after a reset too
again = all(gemm_ok(s) and gemm_ok(s) for s in streams[:ok:16])
rc_reset = L.tnr_gemm_queue_reset(streams[0].cuda_stream)
after_reset = gemm_ok(streams[0])
rc_unbound = L.tnr_gemm_queue_reset(streams[-1].cuda_stream)     # the refused stream has no counters: nothing to reset, still refused
# ONE table for both builds of the library (bf16 / fp16 entry points) and both persistent kernels (NT, weight gradient): a
# stream bound by the bf16 NT launches above runs the fp16 NT kernel and the weight-gradient kernels of both builds on the same
# counters, and the refused stream is refused by all of them.  tnr_gemm_queue_reset then reaches whichever kernel ran last.
def others_ok(stream):
    good = True
    with torch.cuda.stream(stream):
        c16 = torch.zeros((M, N), device=dev)
        T.call("tnr_gemm_nt_f16", a.to(torch.float16), K, b.to(torch.float16), K, c16, N, M, N, K, None, None, 0, None, 0, T.EPI_OUTF32)
        Mw, Nw, Kw = 1024, 512, 256
        for sfx, td in (("", torch.bfloat16), ("_f16", torch.float16)):
            dy = torch.from_numpy(rs.randint(-2, 3, (Mw, Nw)).astype(np.float32)).to(dev)
            x = torch.from_numpy(rs.randint(-2, 3, (Mw, Kw)).astype(np.float32)).to(dev)
            dw = torch.zeros((Nw, Kw), device=dev)
            ws = torch.zeros(Nw * Kw * 4, device=dev)
            T.call("tnr_gemm_tn_wgrad" + sfx, dy.to(td), Nw, x.to(td), Kw, dw, Kw, Mw, Nw, Kw, ws, 4, 0)
            good = good and bool(torch.equal(dw, dy.t() @ x))
    stream.synchronize()
    return good and bool(torch.equal(c16, want))


both_builds = all(others_ok(s) and gemm_ok(s) for s in streams[:ok:32])
try:
    others_ok(streams[-1])
    refused_f16 = False
except T.TnrError:
    refused_f16 = True
rc_reset2 = L.tnr_gemm_queue_reset(streams[0].cuda_stream)
after_reset2 = others_ok(streams[0]) and gemm_ok(streams[0])
L.tnr_gemm_set_option(b"pp", 0)                                      # the documented way out: the non-persistent kernel needs no slot
plain_on_refused = gemm_ok(streams[-1])
L.tnr_gemm_set_option(b"pp", 1)
print(json.dumps(dict(ok=ok, refused_at=refused_at, msg=msg, again=again, rc_reset=rc_reset, after_reset=after_reset,
                      rc_unbound=rc_unbound, plain_on_refused=plain_on_refused, both_builds=both_builds, refused_f16=refused_f16,
                      rc_reset2=rc_reset2, after_reset2=after_reset2)))
