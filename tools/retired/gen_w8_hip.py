"""Generator of a hand-scheduled EIGHT-wave NT GEMM tile kernel WITHOUT the ping-pong segments (tools only; round-4 experiment):
the product kernel's tile, wave layout (2 x 4 waves of 128 x 64) and LDS image, but every wave runs a software-pipelined loop -
the fragment reads of the next k half and its LDS-DMA issues sit BETWEEN its own MFMAs (fragments double-buffered: 96 VGPRs +
128 accumulator AGPRs), one barrier per K tile; the two waves of a SIMD (w, w + 4) issue their LDS-DMA in different places of
the loop body, so that one wave's ~70-cycle DMA issue runs beside its partner's MFMAs.  Scaffolding in HIP C++, K loop + plain
store in ONE inline-assembly statement (as tools/w4_proto/gen_w4_hip.py, whose four-wave form pays the DMA issue serially).

    python tools/w4_proto/gen_w8_hip.py [knobs] > /tmp/w8.hip ; hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/w4_proto/libw8asm.so /tmp/w8.hip
Variants (all in one library; only `full` computes C): full | mfma (no reads / DMA inside the loop) | noread | nodma."""
import sys

# bar_at: MFMA of part 1 behind which the K tile's barrier sits; rd_step: MFMAs per fragment read; dmaA* / dmaB*: the MFMA
# positions (part 0 / part 1) behind which wave group A (waves 0-3) / B (waves 4-7) issue their LDS-DMA pieces
KN = dict(bar_at=8, rd1_step=2, rd0_step=1, dmaA0="2,10,18,26", dmaA1="12,18,24,30", dmaB0="6,14,22,30", dmaB1="14,20,26,31", stagger=0)
if len(sys.argv) > 1 and sys.argv[1]:
    for kv in sys.argv[1].split(";"):
        k, v = kv.split("=")
        KN[k] = v if k.startswith("dma") else int(v)
POS = {k: [int(x) for x in str(KN[k]).split(",")] for k in ("dmaA0", "dmaA1", "dmaB0", "dmaB1")}

FRAG_A = lambda s, i: 16 + (s * 8 + i) * 4          # v16..v79
FRAG_B = lambda s, j: 80 + (s * 4 + j) * 4          # v80..v111
ACC = lambda i, j: (i * 4 + j) * 4                  # a[0:127]
vr = lambda b, n=4: "v[%d:%d]" % (b, b + n - 1)
ar = lambda b, n=4: "a[%d:%d]" % (b, b + n - 1)
ORDER = [(i, j) for i in range(8) for j in range(4)]                      # MFMA (i, j) needs A_i and B_0..3
READ_SEQ = [("A", 0)] + [("B", j) for j in range(4)] + [("A", i) for i in range(1, 8)]


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


def gen_body(variant, grp):
    """The K loop of wave group `grp` ('A': waves 0-3, 'B': waves 4-7) as a list of instructions."""
    DO_READ = variant in ("full", "nodma")
    DO_DMA = variant in ("full", "noread")
    L = []
    e = L.append
    P0, P1 = POS["dma%s0" % grp], POS["dma%s1" % grp]

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
        """Part 0: 32 MFMAs on k half 0; between them the 12 fragment reads of k half 1 and pieces 4-7 of the NEXT K tile.
        Part 1: 32 MFMAs on k half 1; behind MFMA bar_at: vmcnt(0) + barrier (the next K tile has landed, this stage is free), then
        the 12 reads of the next K tile's k half 0 (other stage) and pieces 0-3 of the K tile after that (into this stage)."""
        nxt = stage ^ 1
        rd = list(READ_SEQ) if DO_READ else []
        dm = list(range(4, 8)) if (DO_DMA and dma_p0) else []
        for n, (i, j) in enumerate(ORDER):
            w = lg.need([("A", 0, i)] + [("B", 0, x) for x in range(4)])
            if w:
                e(w)
            mfma(0, i, j)
            if rd and n % KN["rd1_step"] == 0:
                wq, x = rd.pop(0)
                read(stage, 1, wq, x)
                lg.issue((wq, 1, x))
            if dm and n in P0:
                dma(nxt, dm.pop(0))
        while rd:
            wq, x = rd.pop(0)
            read(stage, 1, wq, x)
            lg.issue((wq, 1, x))
        for q in dm:
            dma(nxt, q)
        rd = list(READ_SEQ) if (DO_READ and read_next) else []
        dm = list(range(0, 4)) if (DO_DMA and dma_p1) else []
        for n, (i, j) in enumerate(ORDER):
            w = lg.need([("A", 1, i)] + [("B", 1, x) for x in range(4)])
            if w:
                e(w)
            mfma(1, i, j)
            if n == KN["bar_at"] and bar:
                e("s_waitcnt vmcnt(0)")
                e("s_barrier")
                e("s_add_u32 %[koff], %[koff], 128")
            if n > KN["bar_at"] and bar:
                if rd and (n - KN["bar_at"] - 1) % KN["rd0_step"] == 0:
                    wq, x = rd.pop(0)
                    read(nxt, 0, wq, x)
                    lg.issue((wq, 0, x))
                if dm and n in P1:
                    dma(stage, dm.pop(0))
        while rd:
            wq, x = rd.pop(0)
            read(nxt, 0, wq, x)
            lg.issue((wq, 0, x))
        for q in dm:
            dma(stage, q)

    # K tile 0 entirely (8 pieces per wave), pieces 0-3 of K tile 1
    e("s_mov_b32 %[koff], 0")
    for q in range(8):
        dma(0, q)
    e("s_mov_b32 %[koff], 128")
    if DO_DMA:
        for q in range(4):
            dma(1, q)
        e("s_waitcnt vmcnt(4)")
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
    e("s_cbranch_scc1 .Ltail%s_%%=" % grp)
    e(".Lloop%s_%%=:" % grp)
    lgl = Lgkm(lg.q)
    ktile(0, lgl)
    ktile(1, lgl)
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_lg_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lloop%s_%%=" % grp)
    e(".Ltail%s_%%=:" % grp)
    lgt = Lgkm(lg.q)
    ktile(0, lgt, dma_p1=False)
    ktile(1, lgt, dma_p0=False, dma_p1=False, read_next=False, bar=False)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return L


def gen(variant):
    L = []
    e = L.append
    for r in range(128):
        e("v_accvgpr_write_b32 a%d, 0" % r)
    # waves 4-7 (the second wave of every SIMD) take their own copy of the loop, with the LDS-DMA issues placed elsewhere
    e("s_cmp_ge_u32 %[wave], 4")
    e("s_cbranch_scc1 .LgrpB_%=")
    L += gen_body(variant, "A")
    e("s_branch .Ljoin_%=")
    e(".LgrpB_%=:")
    L += gen_body(variant, "B")
    e(".Ljoin_%=:")
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    if variant == "full":
        # plain store: lane holds C[row + 16 i][col + 16 j .. + 3] (cplo / cphi point at i = j = 0)
        for i in range(8):
            e("v_add_u32 v112, %d, %%[row]" % (16 * i))
            e("v_cmp_gt_u32 vcc, %[msz], v112")
            e("s_and_saveexec_b64 %[save], vcc")
            if i:
                e("s_mul_i32 %%[soff], %%[ldc16], %d" % i)
            else:
                e("s_mov_b32 %[soff], 0")
            e("v_mov_b32 v113, %[soff]")
            e("v_add_co_u32 v114, vcc, %[cplo], v113")
            e("v_addc_co_u32 v115, vcc, 0, %[cphi], vcc")
            for j in range(4):
                b = ACC(i, j)
                for r in range(4):
                    e("v_accvgpr_read_b32 v%d, a%d" % (116 + r, b + r))
                e("s_nop 1")
                e("v_cvt_pk_f16_f32 v120, v116, v117")
                e("v_cvt_pk_f16_f32 v121, v118, v119")
                e("global_store_dwordx2 v[114:115], v[120:121], off offset:%d" % (32 * j))
                e("s_nop 1")
            e("s_mov_b64 exec, %[save]")
    body = "\n".join('        "%s\\n\\t"' % l for l in L)
    clob = ", ".join(['"v%d"' % r for r in range(16, 128)] + ['"a%d"' % r for r in range(128)] + ['"vcc"', '"scc"', '"memory"'])
    return '''
extern "C" __global__ __launch_bounds__(512, 2) void w8_%(v)s(const char* A, const char* B, char* C, int lda, int ldb, int ldc, int M, int nk, int nbn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x %% nbn;
    // wave w stages pieces 8 (w & 1) .. + 7 (8 rows each) of half tile w >> 1 (A rows 0-127, A rows 128-255, B rows 0-127, B rows 128-255)
    const int h = w >> 1;
    const bool isA = h < 2;
    const int ld = isA ? lda : ldb;
    const int row0 = (isA ? bm : bn) * 256 + (h & 1) * 128 + (w & 1) * 64;
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
    const int ldsd = __builtin_amdgcn_readfirstlane((int)lds0 + h * 16384 + (w & 1) * 8192);
    unsigned foff[2];
    for (int s = 0; s < 2; ++s) foff[s] = (unsigned)((lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4));
    const unsigned abase = lds0 + wm * 16384, bbase = lds0 + 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192;
    const unsigned fa00 = abase + foff[0], fa01 = abase + foff[1], fa10 = fa00 + 65536, fa11 = fa01 + 65536;
    const unsigned fb00 = bbase + foff[0], fb01 = bbase + foff[1], fb10 = fb00 + 65536, fb11 = fb01 + 65536;
    const int row = bm * 256 + wm * 128 + (lane & 15), col = bn * 256 + wn * 64 + 4 * (lane >> 4);
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
          [msz] "s"(M), [cplo] "v"(cplo), [cphi] "v"(cphi), [ldc16] "s"(ldc16), [wave] "s"(w)
        : %(clob)s);
}
''' % dict(v=variant, body=body, clob=clob)


print('''// GENERATED by tools/w4_proto/gen_w8_hip.py (knobs: %s) - do not edit
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
        (void)hipFuncSetAttribute((const void*)w8_full, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w8_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w8_noread, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w8_nodma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    }
    const unsigned grid = (unsigned)(((M + 255) / 256) * (N / 256));
    void (*k)(const char*, const char*, char*, int, int, int, int, int, int) = variant == 0 ? w8_full : variant == 1 ? w8_mfma : variant == 2 ? w8_noread : w8_nodma;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, (hipStream_t)stream, (const char*)A, (const char*)B, (char*)C, lda, ldb, ldc, M, K / 64, N / 256);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}''')
