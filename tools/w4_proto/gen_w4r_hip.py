"""Generator of the hand-scheduled FOUR-wave NT GEMM tile kernel with REGISTER STAGING (tools only; round-4 experiment; the
LDS-DMA form is gen_w4r_hip.py): one wave per SIMD owns 512 registers, so 64 of them can carry a K tile's 16 KB share from
`buffer_load_dwordx4` to `ds_write_b128` - an ordinary load costs the issuing wave a few cycles where a `buffer_load ... lds`
stalls it while the vector-memory path is busy (what made gen_w4r_hip.py 33 % slower than the shipped kernel).  Schedule per K
tile t (guide T14, one register set): piece q of tile t+1 is written to the other LDS stage (`vmcnt(15)`: loads return in
order) and the same registers are re-loaded with piece q of tile t+2 at once, 16 times between the MFMAs before the barrier.
The rest as gen_w4r_hip.py: the kernel's
scaffolding (kernel arguments, tile / wave indices, buffer descriptor, LDS and C addresses) is ordinary HIP C++, the K loop and the
plain-store epilogue are ONE inline-assembly statement with fixed fragment / accumulator registers and an explicit schedule.

    C[M, N] (fp16) = A[M, K] . B[N, K]^T, fp16 operands, fp32 accumulation; 256 x 256 x 64 tile per workgroup, four waves of
    128 x 128 (one per SIMD: a[0:255] accumulators, v[16:143] fragments), both operands staged by LDS-DMA
    (buffer_load_dwordx4 ... lds, source addresses on the SALU only) into the XOR-swizzled lane-linear image of the product kernel,
    two 64 KB stages, one barrier per K tile.  [gen_w4r: the operands go through v[160:223] instead of LDS-DMA.]

    python tools/w4r_proto/gen_w4r_hip.py [knobs] > /tmp/w4.hip ; hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libw4asm.so /tmp/w4.hip
Variants (all in one library; only `full` computes C): full | mfma (no reads / DMA inside the loop) | noread | nodma."""
import sys

KN = dict(a_same=0, w_at=0, r_at=2, l_at=4, period=6, rd0_start=100, rd0_step=2, bar_at=96)
if len(sys.argv) > 1 and sys.argv[1]:
    for kv in sys.argv[1].split(","):
        k, v = kv.split("=")
        KN[k] = int(v)

FRAG_A = lambda s, i: 16 + (s * 8 + i) * 4          # v16..v79
FRAG_B = lambda s, j: 80 + (s * 8 + j) * 4          # v80..v143
ACC = lambda i, j: (i * 8 + j) * 4                  # a[0:255]
vr = lambda b, n=4: "v[%d:%d]" % (b, b + n - 1)
ar = lambda b, n=4: "a[%d:%d]" % (b, b + n - 1)
STG = lambda q: 160 + 4 * q                          # v160..v223: one K tile's 16 pieces of this wave


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


class Vm:
    """outstanding buffer loads of this wave, oldest first (they return in order)"""
    def __init__(self, q=()):
        self.q = list(q)

    def issue(self, tag):
        self.q.append(tag)

    def need(self, tag):
        if tag not in self.q:
            return None
        i = self.q.index(tag)
        n = len(self.q) - 1 - i
        self.q = self.q[i + 1:]
        return "s_waitcnt vmcnt(%d)" % n


def gen(variant):
    DO_READ = variant in ("full", "nodma", "noepi")
    DO_STAGE = variant in ("full", "noread", "noepi")
    L = []
    e = L.append

    def load(q, vm, tile):
        """piece q (8 rows x 128 B) of the K tile the descriptor s[36:39] points at -> v[STG(q)]"""
        e("buffer_load_dwordx4 %s, %%[voff], s[36:39], s%d offen" % (vr(STG(q)), 40 + q))
        vm.issue((tile, q))

    def advance():
        """the descriptor moves on by one K tile (64 elements = 128 bytes); its byte count shrinks with it"""
        e("s_add_u32 s36, s36, 128")
        e("s_addc_u32 s37, s37, 0")
        e("s_sub_u32 s38, s38, 128")
        e("s_cselect_b32 s38, 0, s38")                              # a wave whose rows lie beyond M has no bytes: stays empty

    def store(stage, q, vm, lg, tile):
        w = vm.need((tile, q))
        if w:
            e(w)
        e("ds_write_b128 %%[wb%d], %s offset:%d" % (stage, vr(STG(q)), q * 1024))
        lg.issue(("W", q))

    def read(stage, s, which, idx):
        if which == "A":
            e("ds_read_b128 %s, %%[fa%d%d] offset:%d" % (vr(FRAG_A(s, idx)), stage, s, idx * 2048))
        else:
            e("ds_read_b128 %s, %%[fb%d%d] offset:%d" % (vr(FRAG_B(s, idx)), stage, s, idx * 2048))

    def mfma(s, i, j):
        e("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (ar(ACC(i, j)), vr(FRAG_B(s, j)), vr(FRAG_A(s, i)), ar(ACC(i, j))))

    def ktile(stage, lg, vm, t, do_write=True, do_load=True, read_next=True, bar=True):
        """K tile t read from `stage`.  One side operation per MFMA gap at most: gap 6q writes piece q of tile t+1 into the other
        stage, gap 6q+2 reads a fragment of this tile's second k half, gap 6q+4 re-loads the piece's registers with tile t+2;
        barrier behind gap 96, then the next tile's first-half fragments."""
        nxt = stage ^ 1
        plan = {}
        rd1 = list(READ_SEQ) if DO_READ else []
        for q in range(16):
            if DO_STAGE and do_write:
                plan.setdefault(KN["period"] * q + KN["w_at"], []).append(("W", q))
            if rd1:
                plan.setdefault(KN["period"] * q + KN["r_at"], []).append(("R1",) + rd1.pop(0))
            if DO_STAGE and do_write and do_load:
                plan.setdefault(KN["period"] * q + KN["l_at"], []).append(("L", q))
        rd0 = list(READ_SEQ) if (DO_READ and read_next and bar) else []
        g = KN["rd0_start"]
        while rd0:
            plan.setdefault(min(g, 127), []).append(("R0",) + rd0.pop(0))
            g += KN["rd0_step"]
        seq = [(0, i, j) for (i, j) in ORDER] + [(1, i, j) for (i, j) in ORDER]
        for gap, (s_, i, j) in enumerate(seq):
            w = lg.need([("A", s_, i), ("B", s_, j)])
            if w:
                e(w)
            mfma(s_, i, j)
            if gap == KN["bar_at"] and bar:
                w = lg.need([("W", 15)]) if (DO_STAGE and do_write) else None
                if w:
                    e(w)                                            # this wave's pieces of tile t+1 are in LDS
                e("s_barrier")
                if do_load and DO_STAGE:
                    advance()
            for op in plan.get(gap, []):
                if op[0] == "W":
                    store(nxt, op[1], vm, lg, t + 1)
                elif op[0] == "L":
                    load(op[1], vm, t + 2)
                elif op[0] == "R1":
                    read(stage, 1, op[1], op[2])
                    lg.issue((op[1], 1, op[2]))
                else:
                    read(nxt, 0, op[1], op[2])
                    lg.issue((op[1], 0, op[2]))

    for r in range(256):
        e("v_accvgpr_write_b32 a%d, 0" % r)
    # K tile 0 through the staging registers into stage 0, K tile 1 into the registers behind it
    vm = Vm()
    lg = Lgkm()
    e("s_mov_b64 s[36:37], %[srd]")                                 # low pair / high pair of the descriptor operand
    e("s_mov_b32 s38, %[nrec]")
    e("s_mov_b32 s39, 0x00020000")
    for q in range(16):
        e("s_mul_i32 s%d, %%[ld8], %d" % (40 + q, q))
    for q in range(16):
        load(q, vm, 0)
    advance()
    for q in range(16):
        store(0, q, vm, lg, 0)
        if DO_STAGE:
            load(q, vm, 1)
    if not DO_STAGE:                                                # variants without staging in the loop: both stages filled once
        for q in range(16):
            load(q, vm, 1)
        for q in range(16):
            store(1, q, vm, lg, 1)
    advance()
    e("s_waitcnt lgkmcnt(0)")
    lg = Lgkm()
    e("s_barrier")
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
    lgl, vml = Lgkm(lg.q), Vm([(1, q) for q in range(16)] if DO_STAGE else [])
    ktile(0, lgl, vml, 0)
    vml.q = [(1, q) for (_, q) in vml.q]                            # the next trip's view: what is in flight is "tile t+1" again
    ktile(1, lgl, vml, 0)
    e("s_sub_u32 %[cnt], %[cnt], 1")
    e("s_cmp_lg_u32 %[cnt], 0")
    e("s_cbranch_scc1 .Lloop_%=")
    e(".Ltail_%=:")
    lgt, vmt = Lgkm(lg.q), Vm([(1, q) for q in range(16)] if DO_STAGE else [])
    ktile(0, lgt, vmt, 0, do_load=False)
    ktile(1, lgt, vmt, 1, do_write=False, do_load=False, read_next=False, bar=False)
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
    clob = ", ".join(['"v%d"' % r for r in range(16, 224)] + ['"a%d"' % r for r in range(256)] + ['"s%d"' % r for r in range(36, 56)] + ['"vcc"', '"scc"', '"memory"'])
    return '''
extern "C" __global__ __launch_bounds__(256, 1) void w4r_%(v)s(const char* A, const char* B, char* C, int lda, int ldb, int ldc, int M, int nk, int nbn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x %% nbn;
    const bool isA = w < 2;
    const int ld = isA ? lda : ldb;
    const int row0 = (isA ? (%(asame)s ? 0 : bm) : bn) * 256 + (w & 1) * 128;      // a_same: every tile stages A rows 0-255 (L2-resident: a latency probe, wrong results)
    int left = isA ? M - row0 : 65536;
    left = left < 0 ? 0 : (left > 65536 ? 65536 : left);
    const unsigned long long p = (unsigned long long)(isA ? A : B) + (unsigned long long)row0 * (unsigned)ld;
    const unsigned long long srd = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)p) & 0xffffffffull |
                                   ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu)) << 32);
    const int nrec = __builtin_amdgcn_readfirstlane(left * ld);
    const unsigned voff = (unsigned)(lane >> 3) * (unsigned)ld + (unsigned)((((lane & 7) ^ (lane >> 3))) << 4);
    const unsigned lds0 = (unsigned)(unsigned long long)smem;
    const int ld8 = __builtin_amdgcn_readfirstlane(ld * 8);
    const unsigned wb0 = lds0 + (unsigned)w * 16384u + (unsigned)lane * 16u, wb1 = wb0 + 65536u;
    unsigned foff[2];
    for (int s = 0; s < 2; ++s) foff[s] = (unsigned)((lane & 15) * 128 + ((((4 * s) + (lane >> 4)) ^ (lane & 7)) << 4));
    const unsigned fa00 = lds0 + wm * 16384 + foff[0], fa01 = lds0 + wm * 16384 + foff[1], fa10 = fa00 + 65536, fa11 = fa01 + 65536;
    const unsigned fb00 = lds0 + 32768 + wn * 16384 + foff[0], fb01 = lds0 + 32768 + wn * 16384 + foff[1], fb10 = fb00 + 65536, fb11 = fb01 + 65536;
    const int row = bm * 256 + wm * 128 + (lane & 15), col = bn * 256 + wn * 128 + 4 * (lane >> 4);
    const unsigned long long cp = (unsigned long long)C + (unsigned long long)row * (unsigned)ldc + (unsigned)(col * 2);
    const unsigned cplo = (unsigned)cp, cphi = (unsigned)(cp >> 32);
    const int ldc16 = __builtin_amdgcn_readfirstlane(ldc * 16);
    int cnt, soff;
    unsigned long long save;
    asm volatile(
%(body)s
        : [cnt] "=&s"(cnt), [soff] "=&s"(soff), [save] "=&s"(save)
        : [voff] "v"(voff), [srd] "s"(srd), [nrec] "s"(nrec), [ld8] "s"(ld8), [wb0] "v"(wb0), [wb1] "v"(wb1), [fa00] "v"(fa00), [fa01] "v"(fa01), [fa10] "v"(fa10),
          [fa11] "v"(fa11), [fb00] "v"(fb00), [fb01] "v"(fb01), [fb10] "v"(fb10), [fb11] "v"(fb11), [nk] "s"(nk), [row] "v"(row),
          [msz] "s"(M), [cplo] "v"(cplo), [cphi] "v"(cphi), [ldc16] "s"(ldc16)
        : %(clob)s);
}
''' % dict(v=variant, body=body, clob=clob, asame=KN['a_same'])


print('''// GENERATED by tools/w4r_proto/gen_w4r_hip.py (knobs: %s) - do not edit
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
''' % KN)
for v in ("full", "mfma", "noread", "nodma", "noepi"):
    print(gen(v))
print('''
extern "C" int w4r_launch(int variant, const void* A, const void* B, void* C, int lda, int ldb, int ldc, int M, int N, int K, void* stream) {
    if (M < 1 || (N % 256) || (K % 128) || K < 128) return -1;
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute((const void*)w4r_full, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4r_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4r_noread, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4r_nodma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)w4r_noepi, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    }
    const unsigned grid = (unsigned)(((M + 255) / 256) * (N / 256));
    void (*k)(const char*, const char*, char*, int, int, int, int, int, int) = variant == 0 ? w4r_full : variant == 1 ? w4r_mfma : variant == 2 ? w4r_noread : variant == 3 ? w4r_nodma : w4r_noepi;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 131072, (hipStream_t)stream, (const char*)A, (const char*)B, (char*)C, lda, ldb, ldc, M, K / 64, N / 256);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}''')
