"""Generator of the hand-scheduled FOUR-wave NT GEMM tile kernel (tools only; round-4 experiment), second form: the kernel's
scaffolding (kernel arguments, tile / wave indices, buffer descriptor, LDS and C addresses) is ordinary HIP C++, the K loop and the
plain-store epilogue are ONE inline-assembly statement with fixed fragment / accumulator registers and an explicit schedule.

    C[M, N] (fp16) = A[M, K] . B[N, K]^T, fp16 operands, fp32 accumulation; 256 x 256 x 64 tile per workgroup, four waves of
    128 x 128 (one per SIMD: a[0:255] accumulators, v[16:143] fragments), both operands staged by LDS-DMA
    (buffer_load_dwordx4 ... lds, source addresses on the SALU only) into the XOR-swizzled lane-linear image of the product kernel,
    two 64 KB stages, one barrier per K tile.

    python tools/w4_proto/gen_w4_hip.py [knobs] > /tmp/w4.hip ; hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libw4asm.so /tmp/w4.hip
Variants (all in one library; only `full` computes C): full | mfma (no reads / DMA inside the loop) | noread | nodma."""
import sys

KN = dict(dma_step=4, rd1_step=4, rd0_start=36, rd0_step=2, bar_at=32)
if len(sys.argv) > 1 and sys.argv[1]:
    for kv in sys.argv[1].split(","):
        k, v = kv.split("=")
        KN[k] = int(v)

FRAG_A = lambda s, i: 16 + (s * 8 + i) * 4          # v16..v79
FRAG_B = lambda s, j: 80 + (s * 8 + j) * 4          # v80..v143
ACC = lambda i, j: (i * 8 + j) * 4                  # a[0:255]
vr = lambda b, n=4: "v[%d:%d]" % (b, b + n - 1)
ar = lambda b, n=4: "a[%d:%d]" % (b, b + n - 1)


def shell_order():
    o = []
    for k in range(8):
        for j in range(k):
            o.append((k, j))
        for i in range(k):
            o.append((i, k))
        o.append((k, k))
    return o


ORDER = shell_order()
READ_SEQ = [(w, x) for x in range(8) for w in ("A", "B")]


class Lgkm:
    def __init__(self, q=()):
        self.q = list(q)

    def issue(self, tag):
        self.q.append(tag)

    def need(self, tags):
        last = -1
        for t in tags:
            if t in self.q:
                last = max(last, self.q.index(t))
        if last < 0:
            return None
        n = len(self.q) - 1 - last
        self.q = self.q[last + 1:]
        return "s_waitcnt lgkmcnt(%d)" % min(n, 15)


def gen(variant):
    DO_READ = variant in ("full", "nodma")
    DO_DMA = variant in ("full", "noread")
    L = []
    e = L.append

    def dma(stage, q):
        e("s_add_u32 m0, %%[ldsd], 0x%x" % (stage * 65536 + q * 1024))
        if q:
            e("s_mul_i32 %%[soff], %%[ld8], %d" % q)
            e("s_add_u32 %[soff], %[soff], %[koff]")
            e("buffer_load_dwordx4 %[voff], %[srd], %[soff] offen lds")
        else:
            e("s_nop 0")
            e("buffer_load_dwordx4 %[voff], %[srd], %[koff] offen lds")

    def read(stage, s, which, idx):
        if which == "A":
            e("ds_read_b128 %s, %%[fa%d%d] offset:%d" % (vr(FRAG_A(s, idx)), stage, s, idx * 2048))
        else:
            e("ds_read_b128 %s, %%[fb%d%d] offset:%d" % (vr(FRAG_B(s, idx)), stage, s, idx * 2048))

    def mfma(s, i, j):
        e("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (ar(ACC(i, j)), vr(FRAG_B(s, j)), vr(FRAG_A(s, i)), ar(ACC(i, j))))

    def ktile(stage, lg, dma_p0=True, dma_p1=True, read_next=True, bar=True):
        nxt = stage ^ 1
        rd = list(READ_SEQ) if DO_READ else []
        dm = list(range(8, 16)) if (DO_DMA and dma_p0) else []
        for n, (i, j) in enumerate(ORDER):                         # part 0: k half 0
            w = lg.need([("A", 0, i), ("B", 0, j)])
            if w:
                e(w)
            mfma(0, i, j)
            if rd and n % KN["rd1_step"] == 0:
                wq, x = rd.pop(0)
                read(stage, 1, wq, x)
                lg.issue((wq, 1, x))
            if dm and n % KN["dma_step"] == 1:
                dma(nxt, dm.pop(0))
        while rd:
            wq, x = rd.pop(0)
            read(stage, 1, wq, x)
            lg.issue((wq, 1, x))
        for q in dm:
            dma(nxt, q)
        rd = list(READ_SEQ) if (DO_READ and read_next) else []
        dm = list(range(0, 8)) if (DO_DMA and dma_p1) else []
        for n, (i, j) in enumerate(ORDER):                         # part 1: k half 1
            w = lg.need([("A", 1, i), ("B", 1, j)])
            if w:
                e(w)
            mfma(1, i, j)
            if n == KN["bar_at"] and bar:
                e("s_waitcnt vmcnt(0)")
                e("s_barrier")
                e("s_add_u32 %[koff], %[koff], 128")
            if n > KN["bar_at"] and bar:
                if rd and n >= KN["rd0_start"] and (n - KN["rd0_start"]) % KN["rd0_step"] == 0:
                    wq, x = rd.pop(0)
                    read(nxt, 0, wq, x)
                    lg.issue((wq, 0, x))
                if dm and (n - KN["bar_at"]) % KN["dma_step"] == 1:
                    dma(stage, dm.pop(0))
        while rd:
            wq, x = rd.pop(0)
            read(nxt, 0, wq, x)
            lg.issue((wq, 0, x))
        for q in dm:
            dma(stage, q)

    for r in range(256):
        e("v_accvgpr_write_b32 a%d, 0" % r)
    # K tile 0 entirely, pieces 0-7 of K tile 1
    e("s_mov_b32 %[koff], 0")
    for q in range(16):
        dma(0, q)
    e("s_mov_b32 %[koff], 128")
    if DO_DMA:
        for q in range(8):
            dma(1, q)
        e("s_waitcnt vmcnt(8)")
    else:
        e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    lg = Lgkm()
    for wq, x in READ_SEQ:
        read(0, 0, wq, x)
        lg.issue((wq, 0, x))
    if not DO_READ:
        for wq, x in READ_SEQ:
            read(0, 1, wq, x)
            lg.issue((wq, 1, x))
    e("s_lshr_b32 %[cnt], %[nk], 1")
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_eq_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Ltail_%=")
    e(".Lloop_%=:")
    lgl = Lgkm(lg.q)
    ktile(0, lgl)
    ktile(1, lgl)
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_lg_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lloop_%=")
    e(".Ltail_%=:")
    lgt = Lgkm(lg.q)
    ktile(0, lgt, dma_p1=False)
    ktile(1, lgt, dma_p0=False, dma_p1=False, read_next=False, bar=False)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    if variant == "full":
        # plain store: lane holds C[row + 16 i][col + 16 j .. + 3], row / col = its own (cptr points there for i = j = 0)
        for i in range(8):
            e("v_add_u32 v144, %d, %%[row]" % (16 * i))
            e("v_cmp_gt_u32 vcc, %[msz], v144")
            e("s_and_saveexec_b64 %[save], vcc")
            if i:
                e("s_mul_i32 %%[soff], %%[ldc16], %d" % i)
            else:
                e("s_mov_b32 %[soff], 0")
            e("v_mov_b32 v145, %[soff]")
            e("v_add_co_u32 v146, vcc, %[cplo], v145")
            e("v_addc_co_u32 v147, vcc, 0, %[cphi], vcc")
            for j in range(8):
                b = ACC(i, j)
                for r in range(4):
                    e("v_accvgpr_read_b32 v%d, a%d" % (148 + r, b + r))
                e("s_nop 1")
                e("v_cvt_pk_f16_f32 v152, v148, v149")
                e("v_cvt_pk_f16_f32 v153, v150, v151")
                e("global_store_dwordx2 v[146:147], v[152:153], off offset:%d" % (32 * j))
                e("s_nop 1")
            e("s_mov_b64 exec, %[save]")
    body = "\n".join('        "%s\\n\\t"' % l for l in L)
    clob = ", ".join(['"v%d"' % r for r in range(16, 160)] + ['"a%d"' % r for r in range(256)] + ['"vcc"', '"scc"', '"memory"'])
    return '''
extern "C" __global__ __launch_bounds__(256, 1) void w4_%(v)s(const char* A, const char* B, char* C, int lda, int ldb, int ldc, int M, int nk, int nbn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x %% nbn;
    const bool isA = w < 2;
    const int ld = isA ? lda : ldb;
    const int row0 = (isA ? bm : bn) * 256 + (w & 1) * 128;
    int left = isA ? M - row0 : 65536;
    left = left < 0 ? 0 : (left > 65536 ? 65536 : left);
    const unsigned long long p = (unsigned long long)(isA ? A : B) + (unsigned long long)row0 * (unsigned)ld;
    i32x4 srd;
    srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    srd[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    srd[2] = __builtin_amdgcn_readfirstlane(left * ld);
    srd[3] = 0x00020000;
    const unsigned voff = (unsigned)(lane >> 3) * (unsigned)ld + (unsigned)((((lane & 7) ^ (lane >> 3))) << 4);
    const unsigned lds0 = (unsigned)(unsigned long long)smem;
    const int ld8 = __builtin_amdgcn_readfirstlane(ld * 8);
    const int ldsd = __builtin_amdgcn_readfirstlane((int)lds0 + w * 16384);
    unsigned foff[2];
    for (int s = 0; s < 2; ++s) foff[s] = (unsigned)((lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4));
    const unsigned fa00 = lds0 + wm * 16384 + foff[0], fa01 = lds0 + wm * 16384 + foff[1], fa10 = fa00 + 65536, fa11 = fa01 + 65536;
    const unsigned fb00 = lds0 + 32768 + wn * 16384 + foff[0], fb01 = lds0 + 32768 + wn * 16384 + foff[1], fb10 = fb00 + 65536, fb11 = fb01 + 65536;
    const int row = bm * 256 + wm * 128 + (lane & 15), col = bn * 256 + wn * 128 + 4 * (lane >> 4);
    const unsigned long long cp = (unsigned long long)C + (unsigned long long)row * (unsigned)ldc + (unsigned)(col * 2);
    const unsigned cplo = (unsigned)cp, cphi = (unsigned)(cp >> 32);
    const int ldc16 = __builtin_amdgcn_readfirstlane(ldc * 16);
    int cnt, soff, koff;
    unsigned long long save;
    asm volatile(
%(body)s
        : [cnt] "=&s"(cnt), [soff] "=&s"(soff), [koff] "=&s"(koff), [save] "=&s"(save)
        : [voff] "v"(voff), [srd] "s"(srd), [ld8] "s"(ld8), [ldsd] "s"(ldsd), [fa00] "v"(fa00), [fa01] "v"(fa01), [fa10] "v"(fa10),
          [fa11] "v"(fa11), [fb00] "v"(fb00), [fb01] "v"(fb01), [fb10] "v"(fb10), [fb11] "v"(fb11), [nk] "s"(nk), [row] "v"(row),
          [msz] "s"(M), [cplo] "v"(cplo), [cphi] "v"(cphi), [ldc16] "s"(ldc16)
        : %(clob)s);
}
''' % dict(v=variant, body=body, clob=clob)


print('''// GENERATED by tools/w4_proto/gen_w4_hip.py (knobs: %s) - do not edit
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
''' % KN)
for v in ("full", "mfma", "noread", "nodma"):
    print(gen(v))
print('''
extern "C" int w4_launch(int variant, const void* A, const void* B, void* C, int lda, int ldb, int ldc, int M, int N, int K, void* stream) {
    if (M < 1 || (N % 256) || (K % 128) || K < 128) return -1;
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute((const void*)w4_full, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4_noread, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4_nodma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    }
    const unsigned grid = (unsigned)(((M + 255) / 256) * (N / 256));
    void (*k)(const char*, const char*, char*, int, int, int, int, int, int) = variant == 0 ? w4_full : variant == 1 ? w4_mfma : variant == 2 ? w4_noread : w4_nodma;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 131072, (hipStream_t)stream, (const char*)A, (const char*)B, (char*)C, lda, ldb, ldc, M, K / 64, N / 256);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}''')
