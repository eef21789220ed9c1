"""ctypes binding of libtnr_hip.so (include/tnr_hip.h).  No fallback: a missing library or a
non-zero status raises.  torch supplies device memory and the current HIP stream only."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtnr_hip.so")

BF16, F16, F32 = 0, 1, 2
ROUTE_128, ROUTE_256x128, ROUTE_256, ROUTE_224 = 128, 2128, 256, 224
EPI_BIAS, EPI_GELU, EPI_TANH, EPI_RES, EPI_MULDGELU, EPI_OUTF32, EPI_AUXOUT, EPI_COLSUM = 1, 2, 4, 8, 16, 32, 64, 128

_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

DROP_EMB, DROP_PROB, DROP_ATTN_OUT, DROP_FFN_OUT = 0, 1, 2, 3


class Dropout(_c.Structure):
    """tnr_dropout_t (include/tnr_hip.h): one dropout site of one forward pass; pass None for "no dropout"."""
    _fields_ = [("seed", _c.c_uint64), ("site", _c.c_uint32), ("call", _c.c_uint32), ("p", _c.c_double)]

    @classmethod
    def site_of(cls, p, seed, kind, layer, call):
        if p is None or p <= 0.0:
            return None
        return cls(int(seed) & 0xFFFFFFFFFFFFFFFF, int(kind) | (int(layer) << 8), int(call) & 0xFFFFFFFF, float(p))


_D = _c.POINTER(Dropout)


class SgemmProblem(_c.Structure):
    """tnr_sgemm_problem_t (include/tnr_hip.h): one member of a tnr_sgemm_group launch."""
    _fields_ = [("A", _P), ("a_rs", _L), ("a_cs", _L), ("sA", _L), ("B", _P), ("b_rs", _L), ("b_cs", _L), ("sB", _L),
                ("C", _P), ("ldc", _L), ("sC", _L), ("bias", _P), ("sBias", _L), ("M", _L), ("N", _L), ("K", _L),
                ("batch", _I), ("alpha", _F), ("beta", _F), ("ksplit", _I), ("part", _P)]


def sgemm_group(problems):
    """problems: list of dicts with the keyword names of SgemmProblem (tensors or None for the pointers)."""
    arr = (SgemmProblem * len(problems))()
    for i, q in enumerate(problems):
        for k, _ in SgemmProblem._fields_:
            v = q.get(k, 0)
            setattr(arr[i], k, _ptr(v) if (isinstance(v, torch.Tensor) or v is None) else v)
    ev = _all_begin()
    rc = lib().tnr_sgemm_group(arr, len(problems), stream())
    _all_end(ev, "tnr_sgemm_group")
    if rc != 0:
        raise TnrError("tnr_sgemm_group failed (%d): %s" % (rc, lib().tnr_last_error().decode()))

class WgradProblem(_c.Structure):
    """tnr_wgrad_problem_t (include/tnr_hip.h): one member of a tnr_gemm_tn_wgrad_group launch."""
    _fields_ = [("dY", _P), ("lddy", _L), ("X", _P), ("ldx", _L), ("dW", _P), ("lddw", _L), ("M", _L), ("N", _L), ("K", _L),
                ("ws", _P), ("splits", _I), ("accumulate", _I), ("out_scale", _F)]


def wgrad_group(problems, f16=False):
    """problems: list of dicts with the field names of WgradProblem (tensors for the pointers); <= 4."""
    arr = (WgradProblem * len(problems))()
    for i, q in enumerate(problems):
        for k, _ in WgradProblem._fields_:
            v = q[k]
            setattr(arr[i], k, _ptr(v) if isinstance(v, torch.Tensor) else v)
    name = "tnr_gemm_tn_wgrad_group" + ("_f16" if f16 else "")
    ev = _all_begin()
    rc = getattr(lib(), name)(arr, len(problems), stream())
    _all_end(ev, name)
    if rc != 0:
        raise TnrError("%s failed (%d): %s" % (name, rc, lib().tnr_last_error().decode()))


def source_sha16():
    """Hash of the library's SOURCES (csrc/*.hip, *.h, *.cpp, the Makefile, include/tnr_hip.h): what ties a committed rocprofv3 PMC
    summary (profiles/*_pmc.json) to the library bench.py is running - a rebuild of unchanged sources gives another binary hash
    (round 3 review) but the same kernels."""
    import glob, hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   glob.glob(os.path.join(here, "csrc", "*.cpp")) + [os.path.join(here, "csrc", "Makefile"),
                                                                     os.path.join(os.path.dirname(here), "include", "tnr_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# name -> argument types (return type int unless listed in _RET)
_SIG = {
    "tnr_version": [],
    "tnr_relpos_table": [_P, _I, _I, _P, _P],
    "tnr_embed_ln_fwd": [_P, _L, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P],
    "tnr_embed_ln_fwd_indexed": [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P],
    "tnr_gemm_nt": [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P, _L, _P, _L, _I, _P],
    "tnr_gemm_nt_ex": [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P, _L, _P, _L, _I, _P, _P],
    "tnr_gemm_colsum_rows": [_L],
    "tnr_gemm_nt_route": [_L, _L, _L, _I],
    "tnr_gemm_nt_plan": [_L, _L, _I, _I, _P, _P, _P],
    "tnr_gemm_queue_reset": [_P],
    "tnr_gemm_clock_stamps": [_P, _L],
    "tnr_comm_unique_id": [_P],
    "tnr_comm_init": [_P, _I, _I, _c.POINTER(_P)],
    "tnr_comm_world": [_P, _c.POINTER(_I), _c.POINTER(_I)],
    "tnr_comm_broadcast": [_P, _P, _L, _I, _P],
    "tnr_comm_allreduce_avg": [_P, _P, _L, _I, _P],
    "tnr_comm_reduce_scatter_allgather": [_P, _P, _P, _L, _I, _P],
    "tnr_comm_destroy": [_P],
    "tnr_gemm_set_option": [_c.c_char_p, _I],
    "tnr_gemm_tn_wgrad": [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _I, _I, _P],
    "tnr_gemm_tn_wgrad_ex": [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _I, _I, _F, _P],
    "tnr_gemm_tn_ws_elems": [_L, _L, _I],
    "tnr_attpool_long_ws_elems": [_L, _I, _I, _I, _L],
    "tnr_ln_fwd": [_P, _P, _P, _F, _P, _P, _L, _I, _P],
    "tnr_ln_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P],
    "tnr_ln_bwd_part_elems": [_L, _I],
    "tnr_ln_bwd_blocks": [_L],
    "tnr_pool_fwd": [_P, _P, _L, _I, _I, _I, _P],
    "tnr_pool_bwd": [_P, _P, _L, _I, _I, _I, _P],
    "tnr_attn_l32_fwd": [_P, _P, _P, _P, _L, _I, _I, _P],
    "tnr_attn_l32_bwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "tnr_attn_long_fwd": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "tnr_attn_long_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "tnr_colsum": [_P, _L, _I, _L, _L, _P, _P, _I, _P],
    "tnr_colsum_batched": [_P, _L, _L, _I, _L, _L, _I, _P, _P, _I, _P],
    "tnr_colsum_part_elems": [_L, _L],
    "tnr_attpool_fwd": [_P, _P, _L, _P, _P, _I, _P, _P, _P, _L, _I, _I, _P],
    "tnr_attpool_bwd": [_P, _P, _L, _P, _I, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _I, _I, _P],
    "tnr_attpool_fwd_long": [_P, _P, _L, _P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P],
    "tnr_attpool_bwd_long": [_P, _P, _L, _P, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _I, _I, _P],
    "tnr_sgemm": [_P, _L, _L, _L, _P, _P, _L, _L, _L, _P, _L, _L, _P, _L, _L, _L, _L, _I, _F, _F, _I, _P, _P],
    "tnr_sgemm_group": [_c.POINTER(SgemmProblem), _I, _P],
    "tnr_gemm_tn_wgrad_group": [_c.POINTER(WgradProblem), _I, _P],
    "tnr_concat_i32": [_P, _L, _P, _L, _P, _P],
    "tnr_gather_rows": [_P, _L, _P, _L, _I, _I, _P, _L, _L, _P],
    "tnr_segment_sum_rows": [_P, _P, _P, _L, _I, _P, _P],
    "tnr_user_score_fwd": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tnr_user_bwd_pre": [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "tnr_user_bwd_post": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "tnr_user_bwd_part_stride": [_I, _I],
    "tnr_user_blend_fwd": [_P, _L, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "tnr_user_blend_bwd": [_P, _P, _P, _I, _P, _P, _L, _I, _I, _I, _P],
    "tnr_nrms_attn_fwd": [_P, _P, _I, _P, _L, _I, _I, _I, _I, _P],
    "tnr_nrms_attn_bwd": [_P, _P, _I, _P, _P, _I, _I, _I, _P],
    "tnr_score_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "tnr_kd_score_loss": [_P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _I, _P],
    "tnr_kd_embed_loss": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "tnr_reduce_rows": [_P, _L, _L, _L, _P, _I, _P],
    "tnr_reduce_multi": [_P, _I, _P],
    "tnr_dropout_mask": [_D, _L, _L, _P, _P],
    "tnr_dropout_mask_probs": [_D, _L, _I, _I, _P, _P],
    "tnr_scale_inplace": [_P, _L, _F, _P],
    "tnr_amsgrad_step": [_P, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P],
    "tnr_amsgrad_step_guarded": [_P, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P, _c.c_uint, _c.c_uint, _P],
    "tnr_grad_nonfinite": [_P, _L, _P, _c.c_uint, _P],
    "tnr_grad_nonfinite_scan": [_P, _L, _P, _c.c_uint, _P],
    "tnr_grad_nonfinite_commit": [_P, _c.c_uint, _P],
    "tnr_refresh_shadows": [_P, _I, _L, _P, _P],
    "tnr_cast_f32_to_bf16": [_P, _P, _L, _P],
    "tnr_cast_bf16_to_f32": [_P, _P, _L, _P],
}
# entry points that exist twice: bf16 (plain name) and fp16 (suffix _f16)
TYPED = ["tnr_embed_ln_fwd", "tnr_embed_ln_fwd_indexed", "tnr_gemm_nt", "tnr_gemm_nt_ex", "tnr_gemm_colsum_rows", "tnr_gemm_nt_route",
         "tnr_gemm_tn_wgrad", "tnr_gemm_tn_wgrad_ex", "tnr_gemm_tn_wgrad_group", "tnr_gemm_tn_ws_elems", "tnr_ln_fwd", "tnr_ln_bwd", "tnr_attn_l32_fwd", "tnr_attn_l32_bwd", "tnr_attn_long_fwd", "tnr_attn_long_bwd",
         "tnr_colsum", "tnr_colsum_batched", "tnr_attpool_fwd", "tnr_attpool_bwd", "tnr_attpool_fwd_long", "tnr_attpool_bwd_long", "tnr_attpool_long_ws_elems", "tnr_refresh_shadows",
         "tnr_cast_f32_to_bf16", "tnr_cast_bf16_to_f32", "tnr_pool_fwd", "tnr_pool_bwd"]
# *_do: the same with a tnr_dropout_t* in front of the stream (tnr_ln_bwd_do: the masked second output first)
for _n in ("tnr_embed_ln_fwd", "tnr_embed_ln_fwd_indexed", "tnr_attn_l32_fwd", "tnr_attn_l32_bwd", "tnr_attn_long_fwd", "tnr_attn_long_bwd"):
    _SIG[_n + "_do"] = _SIG[_n][:-1] + [_D] + ([_P] if "embed" in _n else []) + [_P]      # embeddings: + pos_ids
    TYPED.append(_n + "_do")
_SIG["tnr_gemm_nt_do"] = _SIG["tnr_gemm_nt_ex"][:-1] + [_D, _P]
_SIG["tnr_ln_bwd_do"] = _SIG["tnr_ln_bwd"][:-1] + [_P, _D, _P]
TYPED += ["tnr_gemm_nt_do", "tnr_ln_bwd_do"]
for _n in TYPED:
    _SIG[_n + "_f16"] = _SIG[_n]
_RET = {"tnr_attpool_long_ws_elems": _L, "tnr_attpool_long_ws_elems_f16": _L, "tnr_gemm_tn_ws_elems": _L, "tnr_gemm_tn_ws_elems_f16": _L, "tnr_gemm_colsum_rows_f16": _L, "tnr_gemm_colsum_rows": _L, "tnr_ln_bwd_part_elems": _L, "tnr_ln_bwd_blocks": _L, "tnr_colsum_part_elems": _L,
        "tnr_user_bwd_part_stride": _L}
EXPORTS = sorted(_SIG) + ["tnr_last_error"]

_lib = None


class TnrError(RuntimeError):
    pass


def lib():
    """Load libtnr_hip.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TnrError("libtnr_hip.so not built: run `make -C tiny-newsrec_amd/csrc` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in _SIG.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RET.get(name, _I)
        L.tnr_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _conv(a):
    if isinstance(a, Dropout):
        return ctypes.byref(a)
    return _ptr(a) if (isinstance(a, torch.Tensor) or a is None) else a


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """torch's current stream on the current device as a raw hipStream_t (~0.2 us through the C entry point, ~1.5 us through the
    Stream object; a step makes ~85 calls)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# name -> list of (start_event, end_event, work) filled while a name is in TIMED (bench.py's roofline leg):
# events are recorded on torch's current stream, which is the stream the kernel is launched on.
TIMED = {}
# a list -> EVERY call is bracketed and recorded as (start_event, end_event, entry point name) (bench.py's step_breakdown_ms: a few
# extra steps AFTER the timed region, so the timed region carries none of it)
TIMED_ALL = None


def _all_begin():
    if TIMED_ALL is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _all_end(e0, name):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        TIMED_ALL.append((e0, e1, name))


def _work(name, conv):
    if name.startswith("tnr_gemm_nt"):
        return 2.0 * conv[6] * conv[7] * conv[8]
    if name.startswith("tnr_gemm_tn_wgrad"):
        return 2.0 * conv[6] * conv[7] * conv[8]
    return 0.0


def call(name, *args):
    """Invoke an int-status entry point on torch's current stream (appended automatically)."""
    L = lib()
    conv = [_conv(a) for a in args]
    rec = TIMED.get(name)
    if rec is not None or TIMED_ALL is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(L, name)(*conv, stream())
    if rec is not None:
        e1.record()
        rec.append((e0, e1, _work(name, conv), (conv[6], conv[7], conv[8], conv[14]) if "gemm_nt" in name else None))
    elif TIMED_ALL is not None:
        e1.record()
        TIMED_ALL.append((e0, e1, name))
    if rc != 0:
        raise TnrError("%s failed (%d): %s" % (name, rc, L.tnr_last_error().decode()))


def query(name, *args):
    return getattr(lib(), name)(*args)
