"""Generates tiny-newsrec_amd/csrc/gemm_nt_pp2.inc: the hand-scheduled K loop of the persistent ping-pong NT GEMM ("pp2").

    python tools/gen_pp2_inc.py > tiny-newsrec_amd/csrc/gemm_nt_pp2.inc          (the .inc is committed; this is its source)

What pp2 changes against the compiler-scheduled loop of gemm_nt_pp_kernel (same tile, wave layout, LDS image, stage ring, ping-pong
of the two wave groups of a SIMD, quadrant order, B register pairs swapping roles every K tile - and therefore the same MFMAs on
the same operands in the same order: bit-identical results): the fragment reads are SOFTWARE-PIPELINED into the MFMA segments.
All fragments of a K tile live in registers (96 VGPRs: A lo, A hi, two B pairs) and each is read one or two segments before the
segment that consumes it, between that earlier segment's MFMAs:
     M0(t): B hi(t) [4 ds_read_b128], A hi(t) [2]    M1(t): A hi(t) [rest]    M2(t): A lo(t+1) [6]    M3(t): A lo(t+1) [2], B lo(t+1) [4]
so a LOAD segment holds nothing but LDS-DMA issues (buffer_load_dwordx4 ... lds: the source address is a per-lane VGPR offset computed
once per tile + the K offset in an SGPR - no per-issue VALU address arithmetic) and a wave never waits for its own fragment reads
in front of its MFMAs.  LDS-DMA of K tile t (every wave: its two 8-row pieces of each half tile A0 A1 B0 B1 = P[0:8]):
     group 0:  L0: P(t+1)[4:8]   L2: vmcnt(0)   L3: P(t+2)[0:4]          group 1:  L1: vmcnt(0)   L2: P(t+2)[0:4]   L3: P(t+2)[4:8]
A stage is free once M1 of its K tile has passed in both groups; a K tile has landed four intervals after its last piece went out;
only vmcnt(0) waits are used, so a wave without A pieces (224-row tiles: wave 7) or with an epilogue's stores in flight needs no
special case.  Measured (tools/w4_proto/gen_pp2_hip.py, the stand-alone form): K loop 1.49 us per K step against 1.85 (N = 3072,
K = 768), 1.56 / 1.84 (N = 2304), 1.63 / 1.84 (N = 768, K = 768); EXPERIMENTS.md round 4.
Registers: v[128:255] accumulators (returned to C++ as eight f32x16 through physical-register constraints; in a[0:127] the
compiler copied all of them to VGPRs in front of the epilogue and spilled around them), v[32:127] fragments,
v[28:31] the fragment base addresses (flipped between the stages once per K tile), everything else operands."""
import sys

FRAG_A = lambda i, s: 32 + (i * 2 + s) * 4          # v32..v95   (i 0-3 lo, 4-7 hi)
BPAIR = lambda p, jj, s: 96 + ((p * 2 + jj) * 2 + s) * 4   # v96..v127
ACC = lambda i, j: 128 + (i * 4 + j) * 4            # v[128:255]
vr = lambda b, n=4: "v[%d:%d]" % (b, b + n - 1)
ar = lambda b, n=4: "v[%d:%d]" % (b, b + n - 1)
PIECES = [(h, q) for h in range(4) for q in range(2)]
FA = ["v28", "v29"]                                  # A fragment base, k half 0 / 1
FB = ["v30", "v31"]


def dma_lines(stage, pcs, lbl):
    """LDS-DMA of the pieces `pcs` of the K tile at byte offset %[koff]; the A pieces are skipped by a wave that has none."""
    L = []
    a = [pc for pc in pcs if pc[0] < 2]
    b = [pc for pc in pcs if pc[0] >= 2]
    if a:
        L.append("s_bitcmp1_b32 %[flags], 1")                      # bit 1: this wave stages A pieces
        L.append("s_cbranch_scc0 .Lna%s_%%=" % lbl)
        for h, q in a:
            L.append("s_add_u32 m0, %%[ldsd], 0x%x" % (stage * 65536 + h * 16384 + q * 1024))
            L.append("s_nop 0")
            L.append("buffer_load_dwordx4 %%[va%d%d], %%[srdA], %%[koff] offen lds" % (h, q))
        L.append(".Lna%s_%%=:" % lbl)
    for h, q in b:
        L.append("s_add_u32 m0, %%[ldsd], 0x%x" % (stage * 65536 + h * 16384 + q * 1024))
        L.append("s_nop 0")
        L.append("buffer_load_dwordx4 %%[vb%d%d], %%[srdB], %%[koff] offen lds" % (h - 2, q))
    return L


def gen_prologue():
    """The first LDS-DMA of a tile's K loop (issued before the previous tile's epilogue): K tile 0 entirely; of K tile 1 group 0
    its first four pieces, group 1 all eight."""
    L = ["s_mov_b32 %[koff], 0"]
    L += dma_lines(0, PIECES, "p0")
    L.append("s_mov_b32 %[koff], 128")
    L += dma_lines(1, PIECES[0:4], "p1")
    L.append("s_bitcmp1_b32 %[flags], 2")                         # bit 2: wave group 1
    L.append("s_cbranch_scc0 .Lpg0_%=")
    L += dma_lines(1, PIECES[4:8], "p2")
    L.append(".Lpg0_%=:")
    return L


def gen_group(grp, MI):
    IHI = MI - 4
    L = []
    e = L.append
    uid = [0]

    def lbl():
        uid[0] += 1
        return "g%dn%d" % (grp, uid[0])

    def seg_end():
        e("s_barrier")
        e("s_setprio 1")

    def mfma_end():
        e("s_waitcnt lgkmcnt(0)")
        e("s_setprio 0")
        e("s_barrier")

    def mseg(quads, reads, cond_last=False):
        """MFMAs of one segment with `reads` between them (one every second MFMA).  cond_last: the MFMAs of the tile's last 16-row
        block (i = MI - 1) are skipped in a short tile (flags bit 0 = tall)."""
        rd = list(reads)
        main = [q for q in quads if not (cond_last and q[0] == MI - 1)]
        last = [q for q in quads if (cond_last and q[0] == MI - 1)]
        for n, (i, j, a, b) in enumerate(main):
            e(TNR_MFMA + " %s, %s, %s, %s" % (ar(ACC(i, j)), vr(b), vr(a), ar(ACC(i, j))))
            if rd and n % 2 == 0:
                e(rd.pop(0))
        for r in rd:
            e(r)
        if last:
            lb = lbl()
            e("s_bitcmp1_b32 %[flags], 0")
            e("s_cbranch_scc0 .Lsk%s_%%=" % lb)
            for (i, j, a, b) in last:
                e(TNR_MFMA + " %s, %s, %s, %s" % (ar(ACC(i, j)), vr(b), vr(a), ar(ACC(i, j))))
            e(".Lsk%s_%%=:" % lb)

    def quad(ahi, bhi, lo_pair, hi_pair):
        out = []
        pair = hi_pair if bhi else lo_pair
        rows = range(4, 4 + IHI) if ahi else range(0, 4)
        for s in range(2):
            for i in rows:
                for jj in range(2):
                    out.append((i, 2 * bhi + jj, FRAG_A(i, s), BPAIR(pair, jj, s)))
        return out

    rdA = lambda i, s: "ds_read_b128 %s, %s offset:%d" % (vr(FRAG_A(i, s)), FA[s], i * 2048)
    # B fragment rows are read in the permuted order of nt_epilogue_cols: block j of the wave's 64 columns = rows 32 (j >> 1) + 4 (j & 1) + ...
    rdB = lambda pair, jj, hi, s: "ds_read_b128 %s, %s offset:%d" % (vr(BPAIR(pair, jj, s)), FB[s], (32 * hi + 4 * jj) * 128)

    def flip():
        for r in FA + FB:
            e("v_xor_b32 %s, 0x10000, %s" % (r, r))

    def ktile(par, issue1, issue2, read_next):
        lo, hi = par, par ^ 1
        nxt = par ^ 1
        hi_rows = list(range(4, 4 + IHI))
        # ---- phase 0: quadrant (A lo, B lo) ; reads B hi(t), first A hi block
        if grp == 0 and issue1:
            L.extend(dma_lines(nxt, PIECES[4:8], lbl()))
        seg_end()
        mseg(quad(0, 0, lo, hi), [rdB(hi, jj, 1, s) for s in range(2) for jj in range(2)] + [rdA(hi_rows[0], 0), rdA(hi_rows[0], 1)])
        mfma_end()
        # ---- phase 1: (A lo, B hi) ; reads the other A hi blocks
        if grp == 1 and issue1:
            e("s_waitcnt vmcnt(0)")
        seg_end()
        mseg(quad(0, 1, lo, hi), [rdA(i, s) for i in hi_rows[1:] for s in range(2)])
        mfma_end()
        flip()                                                     # the fragment bases now point at the stage of K tile t + 1
        # ---- phase 2: (A hi, B hi) ; reads A lo(t + 1) blocks 0-2
        if grp == 0 and issue1:
            e("s_waitcnt vmcnt(0)")
        if grp == 1 and issue2:
            e("s_add_u32 %[koff], %[koff], 128")
            L.extend(dma_lines(par, PIECES[0:4], lbl()))
        seg_end()
        mseg(quad(1, 1, lo, hi), [rdA(i, s) for i in (0, 1, 2) for s in range(2)] if read_next else [], cond_last=True)
        mfma_end()
        # ---- phase 3: (A hi, B lo) ; reads A lo(t + 1) block 3, B lo(t + 1) into the pair B hi(t) has left
        if issue2:
            if grp == 0:
                e("s_add_u32 %[koff], %[koff], 128")
                L.extend(dma_lines(par, PIECES[0:4], lbl()))
            else:
                L.extend(dma_lines(par, PIECES[4:8], lbl()))
        elif grp == 0 and issue1:
            e("s_add_u32 %[koff], %[koff], 128")
        seg_end()
        mseg(quad(1, 0, lo, hi), ([rdA(3, 0), rdA(3, 1)] + [rdB(hi, jj, 0, s) for s in range(2) for jj in range(2)]) if read_next else [],
             cond_last=True)
        mfma_end()

    # tile start: everything the prologue issued has landed; fragments A lo(0), B lo(0) -> pair 0
    e("s_mov_b32 %[koff], 128")
    for i in range(4):
        for s in range(2):
            e(rdA(i, s))
    for s in range(2):
        for jj in range(2):
            e(rdB(0, jj, 0, s))
    e("s_waitcnt lgkmcnt(0)")
    if grp == 1:
        e("s_barrier")                                              # the stagger: group 1 runs one interval behind
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
    return L


def gen_kloop(MI):
    L = []
    e = L.append
    for r in range(128):
        e("v_mov_b32 v%d, 0" % (128 + r))
    e("v_mov_b32 v28, %[fa0]")
    e("v_mov_b32 v29, %[fa1]")
    e("v_mov_b32 v30, %[fb0]")
    e("v_mov_b32 v31, %[fb1]")
    e("s_waitcnt vmcnt(0)")                                        # K tile 0 (and what the prologue issued of K tile 1) is in LDS
    e("s_barrier")
    e("s_bitcmp1_b32 %[flags], 2")
    e("s_cbranch_scc1 .Lgrp1_%=")
    L += gen_group(0, MI)
    e("s_branch .Ljoin_%=")
    e(".Lgrp1_%=:")
    L += gen_group(1, MI)
    e(".Ljoin_%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    return L


def cstr(lines, indent="        "):
    out = []
    for l in lines:
        if TNR_MFMA in l:
            l = l.replace(TNR_MFMA, '" TNR_PP2_MFMA "')
        out.append('%s"%s\\n\\t"' % (indent, l))
    return "\n".join(out)


TNR_MFMA = "@MFMA@"
VOPS = ", ".join('[va%d%d] "v"(va[%d][%d])' % (h, q, h, q) for h in range(2) for q in range(2)) + ", " + \
       ", ".join('[vb%d%d] "v"(vb[%d][%d])' % (h, q, h, q) for h in range(2) for q in range(2))
print('''// GENERATED by tools/gen_pp2_inc.py - do not edit (edit the generator; its docstring describes the schedule)
// Included by gemm.hip inside its anonymous namespace, once per build (TNR_PP2_MFMA = the build's 16x16x32 MFMA mnemonic).

// flags: bit 0 = tall tile (the last 16-row block of both wave groups is computed), bit 1 = this wave stages A pieces, bit 2 = wave group 1
__device__ __forceinline__ void pp2_prologue(const unsigned (&va)[2][2], const unsigned (&vb)[2][2], i32x4_t srdA, i32x4_t srdB, int ldsd, int flags) {
    int koff;
    asm volatile(
%s
        : [koff] "=&s"(koff)
        : %s, [srdA] "s"(srdA), [srdB] "s"(srdB), [ldsd] "s"(ldsd), [flags] "s"(flags)
        : "memory", "scc");
}
''' % (cstr(gen_prologue()), VOPS))
for MI in (8, 7):
    clob = ", ".join(['"v%d"' % r for r in range(28, 128)] + ['"vcc"', '"scc"', '"memory"'])
    outs = ", ".join('"={v[%d:%d]}"(acc[%d])' % (128 + 16 * i, 128 + 16 * i + 15, i) for i in range(8))
    print('''
__device__ __forceinline__ void pp2_kloop_%d(f32x16_t (&acc)[8], const unsigned (&va)[2][2], const unsigned (&vb)[2][2], i32x4_t srdA, i32x4_t srdB,
                                             int ldsd, unsigned fa0, unsigned fa1, unsigned fb0, unsigned fb1, int nk, int flags) {
    int cnt, koff;
    asm volatile(
%s
        : %s, [cnt] "=&s"(cnt), [koff] "=&s"(koff)
        : %s, [srdA] "s"(srdA), [srdB] "s"(srdB), [ldsd] "s"(ldsd), [fa0] "v"(fa0), [fa1] "v"(fa1), [fb0] "v"(fb0), [fb1] "v"(fb1),
          [nk] "s"(nk), [flags] "s"(flags)
        : %s);
}
''' % (MI, cstr(gen_kloop(MI)), outs, VOPS, clob))
