"""Generator of the hand-scheduled ping-pong NT GEMM K loop "pp2" (tools only; round-4 experiment).

The product kernel's tile (256 x 256 x 64), wave layout (2 x 4 waves of 128 x 64), LDS image, two 64 KB stages and ping-pong
structure - the wave groups of a SIMD (rows 0-127 / 128-255) one interval apart, LOAD segment | MFMA segment of 16 MFMAs, four
phases per K tile, quadrants (A lo, B lo), (A lo, B hi), (A hi, B hi), (A hi, B lo) - with ONE change the compiler would not make:
the fragment reads are SOFTWARE-PIPELINED into the MFMA segments.  All fragments of a K tile live in registers (96 VGPRs: A lo,
A hi, two B pairs that swap roles every K tile) and every fragment is read one or two segments before the segment that consumes
it, between that earlier segment's MFMAs:
     M0(t): B hi(t) [4 reads], A hi(t) [2]     M1(t): A hi(t) [6]     M2(t): A lo(t+1) [6]     M3(t): A lo(t+1) [2], B lo(t+1) [4]
so a LOAD segment holds nothing but LDS-DMA issues (buffer_load ... lds, addresses on the SALU) and a wave never waits for its own
fragment reads in front of its MFMAs.  LDS-DMA per K tile t (8 pieces per wave, P = the wave's 2 pieces of each of A0 A1 B0 B1):
     group 0:  L0: P(t+1)[4:8]   L2: vmcnt(0)   L3: P(t+2)[0:4]          group 1:  L1: vmcnt(0)   L2: P(t+2)[0:4]   L3: P(t+2)[4:8]
(a stage is free once M1 of its tile has passed in both groups; a tile has landed four intervals after its last piece was issued.)

    python tools/w4_proto/gen_pp2_hip.py > /tmp/pp2.hip ; hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/w4_proto/libpp2asm.so /tmp/pp2.hip
Variants: full (plain store) | noepi (no stores: the K loop alone, to set beside the product kernel's probe 8)."""
import sys

KN = dict(prio=1, rd_first=0, rd_step=2)
if len(sys.argv) > 1 and sys.argv[1]:
    for kv in sys.argv[1].split(","):
        k, v = kv.split("=")
        KN[k] = int(v)

FRAG_A = lambda i, s: 16 + (i * 2 + s) * 4          # v16..v79   (i 0-3 lo, 4-7 hi)
BPAIR = lambda p, jj, s: 80 + ((p * 2 + jj) * 2 + s) * 4   # v80..v111  pair p (0 / 1), block jj (0 / 1) of the pair
ACC = lambda i, j: (i * 4 + j) * 4                  # a[0:127]
vr = lambda b, n=4: "v[%d:%d]" % (b, b + n - 1)
ar = lambda b, n=4: "a[%d:%d]" % (b, b + n - 1)
PIECES = [(h, q) for h in range(4) for q in range(2)]            # (half tile A0 A1 B0 B1, the wave's piece 0 / 1 of it)


def gen_group(grp, variant):
    """K loop of wave group `grp` (0: waves 0-3, starts at once ; 1: waves 4-7, one interval behind)."""
    L = []
    e = L.append

    def dma(stage, pc):
        h, q = pc
        opnd = "A" if h < 2 else "B"
        e("s_add_u32 m0, %%[ldsd], 0x%x" % (stage * 65536 + h * 16384 + q * 1024))
        k = (h & 1) * 2 + q                                       # 0: +0, 1: + 8 rows, 2: + 128 rows, 3: + 136 rows
        if k == 0:
            e("s_nop 0")
            e("buffer_load_dwordx4 %%[voff%s], %%[srd%s], %%[koff%d] offen lds" % (opnd, opnd, 0))
        else:
            e("s_add_u32 %%[soff], %%[koff0], %%[r%d%s]" % (k, opnd))
            e("buffer_load_dwordx4 %%[voff%s], %%[srd%s], %%[soff] offen lds" % (opnd, opnd))

    # koff0 holds the byte offset of the K tile the NEXT dma() belongs to: set by the caller through set_koff
    def set_koff(expr_tile):
        e("s_lshl_b32 %%[koff0], %s, 7" % expr_tile)

    def seg_end():          # end of a LOAD segment: barrier, then the MFMA segment at raised priority
        e("s_barrier")
        if KN["prio"]:
            e("s_setprio 1")

    def mfma_end():
        e("s_waitcnt lgkmcnt(0)")
        if KN["prio"]:
            e("s_setprio 0")
        e("s_barrier")

    def mseg(quads, reads):
        """16 MFMAs (list of (acc i, j, A reg base, B reg base)) with `reads` (instruction strings) placed between them."""
        rd = list(reads)
        for n, (i, j, a, b) in enumerate(quads):
            e("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (ar(ACC(i, j)), vr(b), vr(a), ar(ACC(i, j))))
            if rd and n >= KN["rd_first"] and (n - KN["rd_first"]) % KN["rd_step"] == 0:
                e(rd.pop(0))
        for r in rd:
            e(r)

    def quad(ilo, jlo, lo_pair, hi_pair):
        """MFMAs of quadrant (A half ilo, B half jlo): i in the A half, j in the B half, s = 0, 1."""
        out = []
        pair = lo_pair if jlo == 0 else hi_pair
        for s in range(2):
            for i in range(4 * ilo, 4 * ilo + 4):
                for jj in range(2):
                    out.append((i, 2 * jlo + jj, FRAG_A(i, s), BPAIR(pair, jj, s)))
        return out

    def rdA(i, s):
        return "ds_read_b128 %s, %%[fa%d] offset:%d" % (vr(FRAG_A(i, s)), s, i * 2048)

    def rdB(pair, jj, jglobal, s):
        return "ds_read_b128 %s, %%[fb%d] offset:%d" % (vr(BPAIR(pair, jj, s)), s, jglobal * 2048)

    def flip():
        for r in ("fa0", "fa1", "fb0", "fb1"):
            e("v_xor_b32 %%[%s], 0x10000, %%[%s]" % (r, r))

    def ktile(par, issue1, issue2, read_next):
        """K tile of parity `par` (stage par; B lo in pair par, B hi in pair par ^ 1).  issue1 / issue2: stage the K tiles t + 1 /
        t + 2 (False near the end of K); read_next: prefetch the fragments of K tile t + 1."""
        lo, hi = par, par ^ 1
        nxt = par ^ 1
        # ---- phase 0
        if grp == 0 and issue1:
            for pc in PIECES[4:8]:
                dma(nxt, pc)
        seg_end()
        mseg(quad(0, 0, lo, hi), [rdB(hi, jj, 2 + jj, s) for s in range(2) for jj in range(2)] + [rdA(4, 0), rdA(4, 1)])
        mfma_end()
        # ---- phase 1
        if grp == 1 and issue1:
            e("s_waitcnt vmcnt(0)")
        seg_end()
        mseg(quad(0, 1, lo, hi), [rdA(i, s) for i in (5, 6, 7) for s in range(2)])
        mfma_end()
        flip()                                                     # the fragment bases now point at the stage of K tile t + 1
        # ---- phase 2
        if grp == 0 and issue1:
            e("s_waitcnt vmcnt(0)")
        if grp == 1 and issue2:
            e("s_add_u32 %[koff0], %[koff0], 128")
            for pc in PIECES[0:4]:
                dma(par, pc)
        seg_end()
        mseg(quad(1, 1, lo, hi), [rdA(i, s) for i in (0, 1, 2) for s in range(2)] if read_next else [])
        mfma_end()
        # ---- phase 3
        if issue2:
            if grp == 0:
                e("s_add_u32 %[koff0], %[koff0], 128")
                for pc in PIECES[0:4]:
                    dma(par, pc)
            else:
                for pc in PIECES[4:8]:
                    dma(par, pc)
        elif grp == 0 and issue1:
            e("s_add_u32 %[koff0], %[koff0], 128")                 # keep koff0 = offset of K tile t + 2 for the next L0
        seg_end()
        mseg(quad(1, 0, lo, hi), ([rdA(3, 0), rdA(3, 1)] + [rdB(hi, jj, jj, s) for s in range(2) for jj in range(2)]) if read_next else [])
        mfma_end()

    # ---------------- prologue: K tile 0 entirely; of K tile 1: group 0 P[0:4], group 1 all; fragments A lo(0), B lo(0) -> pair 0
    e("s_mov_b32 %[koff0], 0")
    for pc in PIECES:
        dma(0, pc)
    e("s_mov_b32 %[koff0], 128")
    for pc in (PIECES[0:4] if grp == 0 else PIECES):
        dma(1, pc)
    e("s_waitcnt vmcnt(%d)" % (4 if grp == 0 else 8))
    e("s_barrier")
    for i in range(4):
        for s in range(2):
            e(rdA(i, s))
    for s in range(2):
        for jj in range(2):
            e(rdB(0, jj, jj, s))
    e("s_waitcnt lgkmcnt(0)")
    # koff0 convention at the top of a K tile t: offset of K tile t + 1
    if grp == 1:
        e("s_barrier")                                              # the stagger: group 1 runs one interval behind
    # ---------------- main loop: pairs of K tiles; the last two K tiles are peeled (nothing left to stage / prefetch)
    e("s_lshr_b32 %[cnt], %[nk], 1")
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_eq_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Ltail%d_%%=" % grp)
    e(".Lloop%d_%%=:" % grp)
    ktile(0, True, True, True)
    ktile(1, True, True, True)
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_lg_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lloop%d_%%=" % grp)
    e(".Ltail%d_%%=:" % grp)
    ktile(0, True, False, True)
    ktile(1, False, False, False)
    if grp == 0:
        e("s_barrier")                                              # group 0 waits for group 1's last MFMA segment
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return L


def gen(variant):
    L = []
    e = L.append
    for r in range(128):
        e("v_accvgpr_write_b32 a%d, 0" % r)
    e("s_cmp_ge_u32 %[wave], 4")
    e("s_cbranch_scc1 .Lgrp1_%=")
    L += gen_group(0, variant)
    e("s_branch .Ljoin_%=")
    e(".Lgrp1_%=:")
    L += gen_group(1, variant)
    e(".Ljoin_%=:")
    e("s_nop 7")
    e("s_nop 7")
    e("s_nop 7")
    if variant == "full":
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
extern "C" __global__ __launch_bounds__(512, 2) void pp2_%(v)s(const char* A, const char* B, char* C, int lda, int ldb, int ldc, int M, int nk, int nbn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x %% nbn;
    // every wave stages pieces 2 w, 2 w + 1 (8 rows each) of each half tile A0 A1 B0 B1; rows past M read as zero
    const int rowA = bm * 256, rowB = bn * 256;
    int leftA = M - rowA;
    leftA = leftA < 0 ? 0 : (leftA > 256 ? 256 : leftA);
    const unsigned long long pa = (unsigned long long)A + (unsigned long long)rowA * (unsigned)lda;
    const unsigned long long pb = (unsigned long long)B + (unsigned long long)rowB * (unsigned)ldb;
    i32x4 srdA, srdB;
    srdA[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa);
    srdA[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu));
    srdA[2] = __builtin_amdgcn_readfirstlane(leftA * lda);
    srdA[3] = 0x00020000;
    srdB[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pb);
    srdB[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(pb >> 32) & 0xffffu));
    srdB[2] = __builtin_amdgcn_readfirstlane(256 * ldb);
    srdB[3] = 0x00020000;
    const unsigned swz = (unsigned)((((lane & 7) ^ (lane >> 3))) << 4);
    const unsigned voffA = (unsigned)(2 * w * 8 + (lane >> 3)) * (unsigned)lda + swz;
    const unsigned voffB = (unsigned)(2 * w * 8 + (lane >> 3)) * (unsigned)ldb + swz;
    const int r1A = __builtin_amdgcn_readfirstlane(lda * 8), r2A = __builtin_amdgcn_readfirstlane(lda * 128), r3A = __builtin_amdgcn_readfirstlane(lda * 136);
    const int r1B = __builtin_amdgcn_readfirstlane(ldb * 8), r2B = __builtin_amdgcn_readfirstlane(ldb * 128), r3B = __builtin_amdgcn_readfirstlane(ldb * 136);
    const unsigned lds0 = (unsigned)(unsigned long long)smem;
    const int ldsd = __builtin_amdgcn_readfirstlane((int)lds0 + 2 * w * 1024);
    unsigned foff[2];
    for (int s = 0; s < 2; ++s) foff[s] = (unsigned)((lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4));
    const unsigned abase = lds0 + wm * 16384, bbase = lds0 + 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192;
    unsigned fa0 = abase + foff[0], fa1 = abase + foff[1], fb0 = bbase + foff[0], fb1 = bbase + foff[1];
    const int row = bm * 256 + wm * 128 + (lane & 15), col = bn * 256 + wn * 64 + 4 * (lane >> 4);
    const unsigned long long cp = (unsigned long long)C + (unsigned long long)row * (unsigned)ldc + (unsigned)(col * 2);
    const unsigned cplo = (unsigned)cp, cphi = (unsigned)(cp >> 32);
    const int ldc16 = __builtin_amdgcn_readfirstlane(ldc * 16);
    int cnt, soff, koff0;
    unsigned long long save;
    asm volatile(
%(body)s
        : [cnt] "=&s"(cnt), [soff] "=&s"(soff), [koff0] "=&s"(koff0), [save] "=&s"(save), [fa0] "+v"(fa0), [fa1] "+v"(fa1), [fb0] "+v"(fb0), [fb1] "+v"(fb1)
        : [voffA] "v"(voffA), [voffB] "v"(voffB), [srdA] "s"(srdA), [srdB] "s"(srdB), [r1A] "s"(r1A), [r2A] "s"(r2A), [r3A] "s"(r3A),
          [r1B] "s"(r1B), [r2B] "s"(r2B), [r3B] "s"(r3B), [ldsd] "s"(ldsd), [nk] "s"(nk), [row] "v"(row),
          [msz] "s"(M), [cplo] "v"(cplo), [cphi] "v"(cphi), [ldc16] "s"(ldc16), [wave] "s"(w)
        : %(clob)s);
}
''' % dict(v=variant, body=body, clob=clob)


print('''// GENERATED by tools/w4_proto/gen_pp2_hip.py (knobs: %s) - do not edit
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
''' % KN)
for v in ("full", "noepi"):
    print(gen(v))
print('''
extern "C" int w4_launch(int variant, const void* A, const void* B, void* C, int lda, int ldb, int ldc, int M, int N, int K, void* stream) {
    if (M < 1 || (N % 256) || (K % 128) || K < 256) return -1;
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute((const void*)pp2_full, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)pp2_noepi, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    }
    const unsigned grid = (unsigned)(((M + 255) / 256) * (N / 256));
    void (*k)(const char*, const char*, char*, int, int, int, int, int, int) = variant == 0 ? pp2_full : pp2_noepi;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, (hipStream_t)stream, (const char*)A, (const char*)B, (char*)C, lda, ldb, ldc, M, K / 64, N / 256);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}''')
