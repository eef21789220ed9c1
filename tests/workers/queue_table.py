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
L.tnr_gemm_set_option(b"pp", 0)                                      # the documented way out: the non-persistent kernel needs no slot
plain_on_refused = gemm_ok(streams[-1])
L.tnr_gemm_set_option(b"pp", 1)
print(json.dumps(dict(ok=ok, refused_at=refused_at, msg=msg, again=again, rc_reset=rc_reset, after_reset=after_reset,
                      rc_unbound=rc_unbound, plain_on_refused=plain_on_refused)))
