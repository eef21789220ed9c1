// Fused self-attention for short titles (L <= 32, head size 64): one wavefront per (sequence, head).
//
// Reference ops replaced (tnlrv3/modeling.py:205-231): transpose_for_scores x3, matmul(Q,K^T), /sqrt(64),
// + attention_mask, + rel_pos, softmax, matmul(P,V), permute+contiguous.
//
// MFMA v_mfma_f32_32x32x16_bf16 on a single 32x32 score tile.  Scores are produced TRANSPOSED
// (S^T = K.Q^T) so a lane owns one query column with its keys in registers: the softmax reductions are
// in-lane plus one lane^32 exchange, and the probability tile is directly the A operand of P.V
// (accumulator-as-operand, k order permuted: element jj of lane half h is key 16s + 8(jj>>2) + 4h + (jj&3)).
// V is staged row-major in a wave-private 4 KB LDS tile and read with ds_read_b64_tr_b16 in that order.
#include "common.h"

namespace {

constexpr float NEG_BIG = -3.0e38f;
// LDS tiles are [32 rows][64 x 16-bit] with a row pitch of TS bytes and TB bytes per tile.  TS = 128 (dense rows) puts rows r
// and r + 2 on the same banks: the 16-byte staging stores of 8 consecutive rows are 4-way conflicted and the two lane halves
// of an accumulator dump (rows r, r + 4) collide; 144 = 36 dwords walks the 64 banks in steps of 4 dwords (attention
// backward counted 15 M conflict cycles per launch, about as many as its busy cycles, with dense rows).
#ifndef TNR_ATTN_TS
#define TNR_ATTN_TS 144
#endif
constexpr int TS = TNR_ATTN_TS;
constexpr int TB = 32 * TS;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// B-operand fragment (k = rows of the LDS tile in accumulator-permuted order, col = 32*ct + (lane&31))
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int s, int ct, int lane) {
    const int g16 = lane >> 4, hh = g16 >> 1, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int row = 16 * s + 4 * hh + q;
    const int col = ct * 32 + 16 * (g16 & 1) + 4 * p;
    bf16x4 v0 = ds_read_tr16(tile + row * TS + col * 2);
    bf16x4 v1 = ds_read_tr16(tile + (row + 8) * TS + col * 2);
    return cat4(v0, v1);
}

// accumulator tile (regs = rows) -> two A-operand fragments of X^T
__device__ __forceinline__ void acc_to_frags(const f32x16& x, bf16x8 (&f)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f[s][jj] = (bf16)x[8 * s + jj];
}

// write an accumulator tile (row = (r&3)+8(r>>2)+4h, col = 32*ct + lane&31) into a [32][64] bf16 LDS tile
__device__ __forceinline__ void acc_to_lds(char* tile, const f32x16& x, int ct, int lane, float scale) {
    const int h = lane >> 5, c = ct * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        *(bf16*)(tile + row * TS + c * 2) = (bf16)(x[r] * scale);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask_add,
                                                       const float* __restrict__ rel, bf16* __restrict__ ctx,
                                                       int64_t n_pairs, int L, int A, TnrDrop drop) {
    __shared__ __attribute__((aligned(16))) char lds[4][TB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t pair = (int64_t)blockIdx.x * 4 + w;
    const bool valid = pair < n_pairs;
    if (!valid) pair = n_pairs - 1;
    const int64_t n = pair / A;
    const int a = (int)(pair - n * A);
    const int HD = A * 64;
    const int64_t ldq = 3 * HD;
    const int row = lane & 31, h = lane >> 5;
    const int rowc = row < L ? row : L - 1;
    char* my = lds[w];

    // additive mask + rel-pos rows of this lane's query FIRST: they are needed right after the score MFMAs, and a load issued
    // there is a second full trip through a saturated memory pipeline (stamps: 7.7 k of a wave's 45 k cycles, tools/attn_stamps.py)
    f32x4 mkv[4], rlv[4];
    {
        const float* relp0 = rel + a * 1024 + row * 32;
        const float* mp0 = mask_add + n * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mkv[g] = *(const f32x4*)(mp0 + 8 * g + 4 * h);
            rlv[g] = *(const f32x4*)(relp0 + 8 * g + 4 * h);
        }
    }
    const bf16* qp = qkv + (n * L + rowc) * ldq + a * 64 + 8 * h;
    bf16x8 qf[4], kf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *(const bf16x8*)(qp + 16 * s);
        kf[s] = *(const bf16x8*)(qp + HD + 16 * s);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = p * 64 + lane, r = idx >> 3, c = idx & 7;
        int rc = r < L ? r : L - 1;
        *(bf16x8*)(my + r * TS + c * 16) = *(const bf16x8*)(qkv + (n * L + rc) * ldq + 2 * HD + a * 64 + c * 8);
    }
    f32x16 st = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) st = TNR_MFMA_32x32x16(kf[s], qf[s], st, 0, 0, 0);

    float mx = NEG_BIG;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 mk = mkv[g], rl = rlv[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = st[4 * g + e] * 0.125f + mk[e] + rl[e];
            st[4 * g + e] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        st[r] = __expf(st[r] - mx);
        sum += st[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] *= inv;
    if (drop.thresh) {                                   // tnlrv3/modeling.py:224: dropout on the normalised probabilities
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float dm[4];
            tnr_drop_prob_row(drop, (uint64_t)pair, 8, row, 8 * g + 4 * h, dm);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[4 * g + e] *= dm[e];
        }
    }
    bf16x8 pf[2];
    acc_to_frags(st, pf);

    f32x16 o[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        o[ct] = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s)
            o[ct] = TNR_MFMA_32x32x16(pf[s], tr_frag(my, s, ct, lane), o[ct], 0, 0, 0);
    }
    acc_to_lds(my, o[0], 0, lane, 1.0f);
    acc_to_lds(my, o[1], 1, lane, 1.0f);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = p * 64 + lane, r = idx >> 3, c = idx & 7;
        if (valid && r < L) *(bf16x8*)(ctx + (n * L + r) * HD + a * 64 + c * 8) = *(const bf16x8*)(my + r * TS + c * 16);
    }
}

// Backward: recompute P in both orientations, then dV = P^T dO, dS = P*(dP - rowsum(dP*P)),
// dQ = dS K / 8, dK = dS^T Q / 8.
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask_add,
                                                       const float* __restrict__ rel, const bf16* __restrict__ dctx,
                                                       bf16* __restrict__ dqkv, float* __restrict__ bias_part,
                                                       int64_t n_pairs, int L, int A, TnrDrop drop) {
    // per wave: K tile, dO tile, Q tile (each 4 KB, row-major [32][64]) + 256 B of row statistics
    __shared__ __attribute__((aligned(16))) char lds[4][3 * TB + 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t pair = (int64_t)blockIdx.x * 4 + w;
    const bool valid = pair < n_pairs;
    if (!valid) pair = n_pairs - 1;
    const int64_t n = pair / A;
    const int a = (int)(pair - n * A);
    const int HD = A * 64;
    const int64_t ldq = 3 * HD;
    const int row = lane & 31, h = lane >> 5;
    const int rowc = row < L ? row : L - 1;
    char* tK = lds[w];
    char* tO = tK + TB;
    char* tQ = tK + 2 * TB;
    float* stat = (float*)(tK + 3 * TB);          // [0..31] = max + log(sum) per query, [32..63] = D per query

    // mask / rel-pos values of both orientations first (see the forward kernel): the query-row view (lane = query) and the
    // key-column view (lane = key; 16 strided rel values) are consumed right after the two MFMA groups below
    f32x4 mkv[4], rlv[4];
    float relcol[16];
    const float* mp = mask_add + n * 32;
    const float mkcol = mp[row];                   // this lane's key (natural orientation)
    {
        const float* relp0 = rel + a * 1024 + row * 32;
        const float* relc0 = rel + a * 1024 + row;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mkv[g] = *(const f32x4*)(mp + 8 * g + 4 * h);
            rlv[g] = *(const f32x4*)(relp0 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) relcol[4 * g + e] = relc0[(8 * g + 4 * h + e) * 32];
        }
    }
    const bf16* qp = qkv + (n * L + rowc) * ldq + a * 64 + 8 * h;
    const bf16* dop = dctx + (n * L + rowc) * HD + a * 64 + 8 * h;
    bf16x8 qf[4], kf[4], vf[4], df[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *(const bf16x8*)(qp + 16 * s);
        kf[s] = *(const bf16x8*)(qp + HD + 16 * s);
        vf[s] = *(const bf16x8*)(qp + 2 * HD + 16 * s);
        df[s] = *(const bf16x8*)(dop + 16 * s);
        if (row >= L) {                            // padded query rows must not contribute to dK / dV
#pragma unroll
            for (int e = 0; e < 8; ++e) df[s][e] = (bf16)0.f;
        }
        int off = row * TS + (16 * s + 8 * h) * 2;
        *(bf16x8*)(tK + off) = kf[s];
        *(bf16x8*)(tO + off) = df[s];
        *(bf16x8*)(tQ + off) = qf[s];
    }

    // ---- transposed orientation: lanes = queries, regs = keys
    f32x16 st = zero16(), dpt = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st = TNR_MFMA_32x32x16(kf[s], qf[s], st, 0, 0, 0);
        dpt = TNR_MFMA_32x32x16(vf[s], df[s], dpt, 0, 0, 0);
    }
    {
        float mx = NEG_BIG;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 mk = mkv[g], rl = rlv[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = st[4 * g + e] * 0.125f + mk[e] + rl[e];
                st[4 * g + e] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - mx);
            sum += st[r];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        float dsum = 0.f;
        if (drop.thresh) {                               // d/dP of (P * mask / (1 - p)) . V : the mask multiplies dP
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dm[4];
                tnr_drop_prob_row(drop, (uint64_t)pair, 8, row, 8 * g + 4 * h, dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) dpt[4 * g + e] *= dm[e];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] *= inv;
            dsum += st[r] * dpt[r];
        }
        dsum += __shfl_xor(dsum, 32, 64);
        if (h == 0) {
            stat[row] = mx + __logf(sum);
            stat[32 + row] = dsum;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = st[r] * (dpt[r] - dsum);      // dS^T
    }
    bf16x8 dstf[2];
    acc_to_frags(st, dstf);
    f32x16 dq[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        dq[ct] = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s)
            dq[ct] = TNR_MFMA_32x32x16(dstf[s], tr_frag(tK, s, ct, lane), dq[ct], 0, 0, 0);
    }

    // ---- natural orientation: lanes = keys, regs = queries
    f32x16 sn = zero16(), dpn = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        sn = TNR_MFMA_32x32x16(qf[s], kf[s], sn, 0, 0, 0);
        dpn = TNR_MFMA_32x32x16(df[s], vf[s], dpn, 0, 0, 0);
    }
    {
        const float mk = mkcol;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 lse = *(const f32x4*)(stat + 8 * g + 4 * h);
            f32x4 dd = *(const f32x4*)(stat + 32 + 8 * g + 4 * h);
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thresh) tnr_drop_prob_col(drop, (uint64_t)pair, 8, 8 * g + 4 * h, row, dm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p = __expf(sn[4 * g + e] * 0.125f + mk + relcol[4 * g + e] - lse[e]);
                sn[4 * g + e] = p * dm[e];                                // what multiplied V in the forward pass
                dpn[4 * g + e] = p * (dpn[4 * g + e] * dm[e] - dd[e]);    // dS
            }
        }
    }
    bf16x8 pnf[2], dsf[2];
    acc_to_frags(sn, pnf);
    acc_to_frags(dpn, dsf);
    f32x16 dv[2], dk[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        dv[ct] = zero16();
        dk[ct] = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            dv[ct] = TNR_MFMA_32x32x16(pnf[s], tr_frag(tO, s, ct, lane), dv[ct], 0, 0, 0);
            dk[ct] = TNR_MFMA_32x32x16(dsf[s], tr_frag(tQ, s, ct, lane), dk[ct], 0, 0, 0);
        }
    }
    // ---- bias gradient partials: column sums over the 32 token rows (rows >= L are exactly zero: padded
    // queries have dO = 0 and padded keys have P = 0), summed from the bf16-rounded values
    if (bias_part != nullptr && valid) {
        float* bp = bias_part + n * ldq + a * 64 + (lane & 31);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float sq = 0.f, sk = 0.f, sv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sq += (float)(bf16)(dq[ct][r] * 0.125f);
                sk += (float)(bf16)(dk[ct][r] * 0.125f);
                sv += (float)(bf16)dv[ct][r];
            }
            sq += __shfl_xor(sq, 32, 64);
            sk += __shfl_xor(sk, 32, 64);
            sv += __shfl_xor(sv, 32, 64);
            if (h == 0) {
                bp[ct * 32] = sq;
                bp[HD + ct * 32] = sk;
                bp[2 * HD + ct * 32] = sv;
            }
        }
    }
    // ---- stage the three gradient tiles (rows = token, cols = head dim) and store coalesced
    acc_to_lds(tK, dq[0], 0, lane, 0.125f);
    acc_to_lds(tK, dq[1], 1, lane, 0.125f);
    acc_to_lds(tO, dk[0], 0, lane, 0.125f);
    acc_to_lds(tO, dk[1], 1, lane, 0.125f);
    acc_to_lds(tQ, dv[0], 0, lane, 1.0f);
    acc_to_lds(tQ, dv[1], 1, lane, 1.0f);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = p * 64 + lane, r = idx >> 3, c = idx & 7;
        if (valid && r < L) {
            bf16* dst = dqkv + (n * L + r) * ldq + a * 64 + c * 8;
            *(bf16x8*)(dst) = *(const bf16x8*)(tK + r * TS + c * 16);
            *(bf16x8*)(dst + HD) = *(const bf16x8*)(tO + r * TS + c * 16);
            *(bf16x8*)(dst + 2 * HD) = *(const bf16x8*)(tQ + r * TS + c * 16);
        }
    }
}

// ================================================================================================
// Long sequences (L <= 512: stage-1 bodies, Post-train_KD.ipynb cell 4): flash-style tiling over 32-key tiles with
// an online softmax.  Same 32x32 MFMA tile machinery as above; one wave per (sequence, head, 32-query tile).
// The output is accumulated TRANSPOSED (O^T = V^T . P^T: rows = head dim in registers, columns = queries on
// lanes) so the per-query rescale factor exp(m_old - m_new) is a per-lane scalar.
// mask_add: (N, Lr) fp32, rel: (A, Lr, Lr) fp32, Lr = roundup(L, 32); lse: (N, A, Lr) fp32 = max + log(sum).
__device__ __forceinline__ bf16x8 ld_frag(const bf16* base, int64_t ld, int row, int h, int s) {
    return *(const bf16x8*)(base + (int64_t)row * ld + 16 * s + 8 * h);
}

// Round 5: the four waves of a workgroup take four CONSECUTIVE query tiles of ONE (sequence, head) and share its key / value
// tiles - each 32-key tile is loaded once per workgroup (256 threads x 16 bytes for K, the same for V, coalesced rows) through
// registers into a double-buffered LDS pair, the next tile's loads in flight under the current tile's MFMAs and softmax.  Before,
// every wave loaded every K fragment (strided 16-byte pieces) and V tile of its (sequence, head) by itself, one after the other
// with nothing in flight: 2.8 x the algorithmic fabric traffic, 2 TB/s, MFMA busy 0.08 (profiles/r05_stage1_24_512_pmc.json).
// Round 6: __launch_bounds__(256, 3).  Left alone the compiler takes 196 registers (2 waves per SIMD); held to 168 it spills two dwords
// and three waves per SIMD hide more of the per-element fp32 work between the MFMAs: 128.5 -> 114.3 us at L = 128, 299 -> 260 us at
// L = 512 (tools/attn_bench.py, interleaved, same results).  Four waves (128 registers) spill 51 dwords inside the loop: 2.6 x slower;
// the dQ pass at three waves spills 50: +45 % - both stay as they are (EXPERIMENTS item 51).
__global__ __launch_bounds__(256, 3) void attn_long_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask_add,
                                                            const float* __restrict__ rel, bf16* __restrict__ ctx,
                                                            float* __restrict__ lse, int64_t n_items, int L, int Lr, int A,
                                                            TnrDrop drop) {
    __shared__ __attribute__((aligned(16))) char sk[2][TB], sv[2][TB], so[4][TB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nqt = Lr >> 5, nqg = (nqt + 3) >> 2;
    const int qg = (int)(blockIdx.x % nqg);
    const int64_t na = blockIdx.x / nqg;                   // (sequence, head): n_items = pairs x query tiles is the host's bound
    (void)n_items;
    const int64_t n = na / A;
    const int a = (int)(na - n * A);
    const int qt_raw = qg * 4 + w;
    const bool valid = qt_raw < nqt;                       // a wave past the last query tile still loads and keeps the barriers
    const int qt = valid ? qt_raw : nqt - 1;
    const int HD = A * 64;
    const int64_t ldq = 3 * HD;
    const int row = lane & 31, h = lane >> 5;
    char* my = so[w];
    const int qi = qt * 32 + row;                          // this lane's query
    const int qic = qi < L ? qi : L - 1;
    const bf16* qbase = qkv + (n * L) * ldq + a * 64;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = ld_frag(qbase, ldq, qic, h, s);
    f32x16 o[2] = {zero16(), zero16()};
    float m_run = NEG_BIG, l_run = 0.f;
    const float* relq = rel + ((int64_t)a * Lr + (qi < Lr ? qi : Lr - 1)) * Lr;
    const float* mp = mask_add + n * Lr;
    const int lr = tid >> 3, lc = tid & 7;                 // this thread's 16 bytes of a [32 keys][64] tile
    bf16x8 kreg, vreg;
    auto gload = [&](int kt) {
        int rc = kt * 32 + lr;
        rc = rc < L ? rc : L - 1;
        kreg = *(const bf16x8*)(qbase + HD + (int64_t)rc * ldq + lc * 8);
        vreg = *(const bf16x8*)(qbase + 2 * HD + (int64_t)rc * ldq + lc * 8);
    };
    // the additive terms of the NEXT key tile (mask of its keys, bias row pieces of this lane's query) are requested a tile ahead as
    // well: asked for behind the score MFMAs they were a round trip to L2 / the MALL in every tile's critical path (the (A, Lr, Lr)
    // bias table does not fit an XCD's L2: 2.2 us per tile for 0.25 us of MFMAs)
    f32x4 mkn[4], rln[4];
    auto aload = [&](int kt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mkn[g] = *(const f32x4*)(mp + kt * 32 + 8 * g + 4 * h);
            rln[g] = *(const f32x4*)(relq + kt * 32 + 8 * g + 4 * h);
        }
    };
    gload(0);
    aload(0);
    for (int kt = 0; kt < nqt; ++kt) {
        const int cur = kt & 1;
        *(bf16x8*)(sk[cur] + lr * TS + lc * 16) = kreg;
        *(bf16x8*)(sv[cur] + lr * TS + lc * 16) = vreg;
        f32x4 mkc[4], rlc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { mkc[g] = mkn[g]; rlc[g] = rln[g]; }
        __syncthreads();                                   // tile kt is in LDS ; everybody has left tile kt - 1 (the other buffer pair)
        if (kt + 1 < nqt) { gload(kt + 1); aload(kt + 1); }
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *(const bf16x8*)(sk[cur] + row * TS + (16 * s + 8 * h) * 2);
        f32x16 st = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) st = TNR_MFMA_32x32x16(kf[s], qf[s], st, 0, 0, 0);
        float mx = NEG_BIG;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 mk = mkc[g], rl = rlc[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = st[4 * g + e] * 0.125f + mk[e] + rl[e];
                st[4 * g + e] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - m_new);
            sum += st[r];
        }
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if (drop.thresh) {                               // the normaliser keeps every key; only what multiplies V is masked
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dm[4];
                tnr_drop_prob_row(drop, (uint64_t)na, Lr >> 2, qi < Lr ? qi : Lr - 1, kt * 32 + 8 * g + 4 * h, dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) st[4 * g + e] *= dm[e];
            }
        }
        bf16x8 pf[2];
        acc_to_frags(st, pf);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
#pragma unroll
            for (int s = 0; s < 2; ++s) o[ct] = TNR_MFMA_32x32x16(tr_frag(sv[cur], s, ct, lane), pf[s], o[ct], 0, 0, 0);
        }
    }
    const float inv = 1.0f / l_run;
    if (valid && h == 0 && qi < Lr) lse[(n * A + a) * Lr + qi] = m_run + __logf(l_run);
    // O^T (rows = head dim, cols = queries) -> LDS tile [query][head dim] -> coalesced rows
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int dd = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            *(bf16*)(my + row * TS + dd * 2) = (bf16)(o[ct][r] * inv);
        }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = p * 64 + lane, r = idx >> 3, c = idx & 7;
        int q = qt * 32 + r;
        if (valid && q < L) *(bf16x8*)(ctx + (n * L + q) * HD + a * 64 + c * 8) = *(const bf16x8*)(my + r * TS + c * 16);
    }
}

// backward pass 1: dQ per (sequence, head, query tile); also writes delta_i = sum_d dO[i][d] O[i][d].
// Round 5, as the forward: the four waves of a workgroup take four consecutive query tiles of ONE (sequence, head) and share its
// K / V tiles (loaded once per workgroup, 16 bytes per thread each, coalesced rows, through registers into a double-buffered LDS
// pair with the next tile in flight); the K tile in LDS serves both as the S^T operand (rows) and, read transposed, as the dQ
// product's.  The same arithmetic in the same order as the one-wave-per-item kernel it replaces: bit-identical results.
__global__ __launch_bounds__(256) void attn_long_bwd_dq_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask_add,
                                                               const float* __restrict__ rel, const bf16* __restrict__ ctx,
                                                               const bf16* __restrict__ dctx, const float* __restrict__ lse,
                                                               float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                               int64_t n_items, int L, int Lr, int A, TnrDrop drop) {
    __shared__ __attribute__((aligned(16))) char lds[4][TB];             // K pair, V pair ; after the loop: one staging tile per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    (void)n_items;
    const int nqt = Lr >> 5, nqg = (nqt + 3) >> 2;
    const int qg = (int)(blockIdx.x % nqg);
    const int64_t na = blockIdx.x / nqg;
    const int64_t n = na / A;
    const int a = (int)(na - n * A);
    const int qt_raw = qg * 4 + w;
    const bool valid = qt_raw < nqt;                       // a wave past the last query tile still loads and keeps the barriers
    const int qt = valid ? qt_raw : nqt - 1;
    const int HD = A * 64;
    const int64_t ldq = 3 * HD;
    const int row = lane & 31, h = lane >> 5;
    const int qi = qt * 32 + row;
    const int qic = qi < L ? qi : L - 1;
    const bf16* qbase = qkv + (n * L) * ldq + a * 64;
    const bf16* obase = ctx + (n * L) * HD + a * 64;
    const bf16* dobase = dctx + (n * L) * HD + a * 64;
    bf16x8 qf[4], df[4];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = ld_frag(qbase, ldq, qic, h, s);
        df[s] = ld_frag(dobase, HD, qic, h, s);
        bf16x8 of = ld_frag(obase, HD, qic, h, s);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)df[s][e] * (float)of[e];
        if (qi >= L) {
#pragma unroll
            for (int e = 0; e < 8; ++e) df[s][e] = (bf16)0.f;
        }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (qi >= L) dl = 0.f;
    const float lse_i = lse[(n * A + a) * Lr + (qi < Lr ? qi : Lr - 1)];
    if (valid && h == 0 && qi < Lr) delta[(n * A + a) * Lr + qi] = dl;
    const float* relq = rel + ((int64_t)a * Lr + (qi < Lr ? qi : Lr - 1)) * Lr;
    const float* mp = mask_add + n * Lr;
    f32x16 dq[2] = {zero16(), zero16()};
    const int lr = tid >> 3, lc = tid & 7;                 // this thread's 16 bytes of a [32 keys][64] tile
    bf16x8 kreg, vreg;
    auto gload = [&](int kt) {
        int rc = kt * 32 + lr;
        rc = rc < L ? rc : L - 1;
        kreg = *(const bf16x8*)(qbase + HD + (int64_t)rc * ldq + lc * 8);
        vreg = *(const bf16x8*)(qbase + 2 * HD + (int64_t)rc * ldq + lc * 8);
    };
    f32x4 mkn[4], rln[4];                                  // the next key tile's additive terms, a tile ahead (see the forward)
    auto aload = [&](int kt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mkn[g] = *(const f32x4*)(mp + kt * 32 + 8 * g + 4 * h);
            rln[g] = *(const f32x4*)(relq + kt * 32 + 8 * g + 4 * h);
        }
    };
    gload(0);
    aload(0);
    for (int kt = 0; kt < nqt; ++kt) {
        char* sk = lds[kt & 1];
        char* sv = lds[2 + (kt & 1)];
        *(bf16x8*)(sk + lr * TS + lc * 16) = kreg;
        *(bf16x8*)(sv + lr * TS + lc * 16) = vreg;
        f32x4 mkc[4], rlc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { mkc[g] = mkn[g]; rlc[g] = rln[g]; }
        __syncthreads();                                   // tile kt is in LDS ; everybody has left tile kt - 1 (the other pair)
        if (kt + 1 < nqt) { gload(kt + 1); aload(kt + 1); }
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *(const bf16x8*)(sk + row * TS + (16 * s + 8 * h) * 2);
            vf[s] = *(const bf16x8*)(sv + row * TS + (16 * s + 8 * h) * 2);
        }
        f32x16 st = zero16(), dpt = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            st = TNR_MFMA_32x32x16(kf[s], qf[s], st, 0, 0, 0);
            dpt = TNR_MFMA_32x32x16(vf[s], df[s], dpt, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 mk = mkc[g], rl = rlc[g];
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thresh) tnr_drop_prob_row(drop, (uint64_t)na, Lr >> 2, qi < Lr ? qi : Lr - 1, kt * 32 + 8 * g + 4 * h, dm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p = __expf(st[4 * g + e] * 0.125f + mk[e] + rl[e] - lse_i);
                st[4 * g + e] = p * (dpt[4 * g + e] * dm[e] - dl);               // dS^T
            }
        }
        bf16x8 dsf[2];
        acc_to_frags(st, dsf);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) dq[ct] = TNR_MFMA_32x32x16(dsf[s], tr_frag(sk, s, ct, lane), dq[ct], 0, 0, 0);
    }
    __syncthreads();                                       // the last tiles have been read: the four buffers become staging tiles
    char* my = lds[w];
    acc_to_lds(my, dq[0], 0, lane, 0.125f);
    acc_to_lds(my, dq[1], 1, lane, 0.125f);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = p * 64 + lane, r = idx >> 3, c = idx & 7;
        int q = qt * 32 + r;
        if (valid && q < L) *(bf16x8*)(dqkv + (n * L + q) * ldq + a * 64 + c * 8) = *(const bf16x8*)(my + r * TS + c * 16);
    }
}

// backward pass 2: dK, dV per (sequence, head, key tile), looping over the query tiles.  Round 5: four consecutive key tiles of one
// (sequence, head) per workgroup share its Q / dO tiles (one coalesced 16-byte load per thread and tile each, double-buffered through
// registers); a wave reads its row fragments and the transposed fragments of the dK / dV products from the shared tiles.  Rows of
// dO past L are zeroed by the loader, rows of Q past L repeat row L - 1, as the per-wave loads did: bit-identical results.
__global__ __launch_bounds__(256, 2) void attn_long_bwd_dkv_kernel(const bf16* __restrict__ qkv, const float* __restrict__ mask_add,
                                                                const float* __restrict__ rel, const bf16* __restrict__ dctx,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16* __restrict__ dqkv, int64_t n_items, int L, int Lr, int A,
                                                                TnrDrop drop) {
    // dO pair, Q pair, this wave's K and V tile (fragments re-read every query tile: held in registers for the whole loop they
    // pushed the kernel over its 256 registers - 84 bytes of scratch per lane) ; after the loop: two staging tiles per wave
    __shared__ __attribute__((aligned(16))) char lds[12][TB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    (void)n_items;
    const int nqt = Lr >> 5, nkg = (nqt + 3) >> 2;
    const int kg = (int)(blockIdx.x % nkg);
    const int64_t na = blockIdx.x / nkg;
    const int64_t n = na / A;
    const int a = (int)(na - n * A);
    const int kt_raw = kg * 4 + w;
    const bool valid = kt_raw < nqt;
    const int kt = valid ? kt_raw : nqt - 1;
    const int HD = A * 64;
    const int64_t ldq = 3 * HD;
    const int row = lane & 31, h = lane >> 5;
    const int kj = kt * 32 + row;                          // this lane's key
    const int kjc = kj < L ? kj : L - 1;
    const bf16* qbase = qkv + (n * L) * ldq + a * 64;
    const bf16* dobase = dctx + (n * L) * HD + a * 64;
    char* const myK = lds[4 + 2 * w];
    char* const myV = lds[5 + 2 * w];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int off = row * TS + (16 * s + 8 * h) * 2;
        *(bf16x8*)(myK + off) = ld_frag(qbase + HD, ldq, kjc, h, s);
        *(bf16x8*)(myV + off) = ld_frag(qbase + 2 * HD, ldq, kjc, h, s);
    }
    const float mk = mask_add[n * Lr + (kj < Lr ? kj : Lr - 1)];
    const float* relc = rel + (int64_t)a * Lr * Lr + (kj < Lr ? kj : Lr - 1);
    const float* lsep = lse + (n * A + a) * Lr;
    const float* dlp = delta + (n * A + a) * Lr;
    f32x16 dv[2] = {zero16(), zero16()}, dk[2] = {zero16(), zero16()};
    const int lr = tid >> 3, lc = tid & 7;                 // this thread's 16 bytes of a [32 queries][64] tile
    bf16x8 qreg, dreg;
    auto gload = [&](int qt) {
        const int q = qt * 32 + lr;
        const int qc = q < L ? q : L - 1;
        qreg = *(const bf16x8*)(qbase + (int64_t)qc * ldq + lc * 8);
        dreg = *(const bf16x8*)(dobase + (int64_t)qc * HD + lc * 8);
        if (q >= L) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dreg[e] = (bf16)0.f;
        }
    };
    gload(0);
    for (int qt = 0; qt < nqt; ++qt) {
        char* tO = lds[qt & 1];
        char* tQ = lds[2 + (qt & 1)];
        *(bf16x8*)(tO + lr * TS + lc * 16) = dreg;
        *(bf16x8*)(tQ + lr * TS + lc * 16) = qreg;
        __syncthreads();                                   // tile qt is in LDS ; everybody has left tile qt - 1 (the other pair)
        if (qt + 1 < nqt) gload(qt + 1);
        bf16x8 qf[4], df[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *(const bf16x8*)(tQ + row * TS + (16 * s + 8 * h) * 2);
            df[s] = *(const bf16x8*)(tO + row * TS + (16 * s + 8 * h) * 2);
        }
        f32x16 sn = zero16(), dpn = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 kfs = *(const bf16x8*)(myK + row * TS + (16 * s + 8 * h) * 2);
            const bf16x8 vfs = *(const bf16x8*)(myV + row * TS + (16 * s + 8 * h) * 2);
            sn = TNR_MFMA_32x32x16(qf[s], kfs, sn, 0, 0, 0);               // S[i][j]: regs = queries, lanes = keys
            dpn = TNR_MFMA_32x32x16(df[s], vfs, dpn, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 ls = *(const f32x4*)(lsep + qt * 32 + 8 * g + 4 * h);
            f32x4 dd = *(const f32x4*)(dlp + qt * 32 + 8 * g + 4 * h);
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.thresh) tnr_drop_prob_col(drop, (uint64_t)na, Lr >> 2, qt * 32 + 8 * g + 4 * h, kj < Lr ? kj : Lr - 1, dm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int q = qt * 32 + 8 * g + 4 * h + e;
                float p = __expf(sn[4 * g + e] * 0.125f + mk + relc[(int64_t)q * Lr] - ls[e]);
                sn[4 * g + e] = p * dm[e];
                dpn[4 * g + e] = p * (dpn[4 * g + e] * dm[e] - dd[e]);
            }
        }
        bf16x8 pnf[2], dsf[2];
        acc_to_frags(sn, pnf);
        acc_to_frags(dpn, dsf);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dv[ct] = TNR_MFMA_32x32x16(pnf[s], tr_frag(tO, s, ct, lane), dv[ct], 0, 0, 0);
                dk[ct] = TNR_MFMA_32x32x16(dsf[s], tr_frag(tQ, s, ct, lane), dk[ct], 0, 0, 0);
            }
    }
    __syncthreads();                                       // the last tiles have been read: buffers 0-7 become staging
    char* sK = lds[2 * w];
    char* sV = lds[2 * w + 1];
    acc_to_lds(sK, dk[0], 0, lane, 0.125f);
    acc_to_lds(sK, dk[1], 1, lane, 0.125f);
    acc_to_lds(sV, dv[0], 0, lane, 1.0f);
    acc_to_lds(sV, dv[1], 1, lane, 1.0f);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int idx = p * 64 + lane, r = idx >> 3, c = idx & 7;
        int j = kt * 32 + r;
        if (valid && j < L) {
            bf16* dst = dqkv + (n * L + j) * ldq + a * 64 + c * 8;
            *(bf16x8*)(dst + HD) = *(const bf16x8*)(sK + r * TS + c * 16);
            *(bf16x8*)(dst + 2 * HD) = *(const bf16x8*)(sV + r * TS + c * 16);
        }
    }
}

}  // namespace

extern "C" int TNR_NAME(tnr_attn_l32_fwd)(const void* qkv, const float* mask_add, const float* rel, void* ctx, int64_t n_seq,
                                int L, int A, void* stream) {
    return TNR_NAME(tnr_attn_l32_fwd_do)(qkv, mask_add, rel, ctx, n_seq, L, A, nullptr, stream);
}
extern "C" int TNR_NAME(tnr_attn_l32_fwd_do)(const void* qkv, const float* mask_add, const float* rel, void* ctx, int64_t n_seq,
                                   int L, int A, const tnr_dropout_t* drop, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_attn_l32_fwd")) return rc;
    TNR_CHECK_ARG(qkv && mask_add && rel && ctx, "tnr_attn_l32_fwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 32 && A >= 1 && n_seq >= 1, "tnr_attn_l32_fwd: need 1<=L<=32");
    TNR_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 16) == 0, "tnr_attn_l32_fwd: 16-byte alignment");
    int64_t pairs = n_seq * A;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)qkv, mask_add, rel, (bf16*)ctx, pairs, L, A, dd);
    TNR_CHECK_LAUNCH("tnr_attn_l32_fwd");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_attn_l32_bwd)(const void* qkv, const float* mask_add, const float* rel, const void* dctx, void* dqkv,
                                float* bias_part, int64_t n_seq, int L, int A, void* stream) {
    return TNR_NAME(tnr_attn_l32_bwd_do)(qkv, mask_add, rel, dctx, dqkv, bias_part, n_seq, L, A, nullptr, stream);
}
extern "C" int TNR_NAME(tnr_attn_l32_bwd_do)(const void* qkv, const float* mask_add, const float* rel, const void* dctx, void* dqkv,
                                   float* bias_part, int64_t n_seq, int L, int A, const tnr_dropout_t* drop, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_attn_l32_bwd")) return rc;
    TNR_CHECK_ARG(qkv && mask_add && rel && dctx && dqkv, "tnr_attn_l32_bwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 32 && A >= 1 && n_seq >= 1, "tnr_attn_l32_bwd: need 1<=L<=32");
    TNR_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)dctx % 16) == 0 && ((uintptr_t)dqkv % 16) == 0,
                  "tnr_attn_l32_bwd: 16-byte alignment");
    int64_t pairs = n_seq * A;
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)qkv, mask_add, rel, (const bf16*)dctx, (bf16*)dqkv, bias_part, pairs, L, A, dd);
    TNR_CHECK_LAUNCH("tnr_attn_l32_bwd");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_attn_long_fwd)(const void* qkv, const float* mask_add, const float* rel, void* ctx, float* lse,
                                           int64_t n_seq, int L, int A, void* stream) {
    return TNR_NAME(tnr_attn_long_fwd_do)(qkv, mask_add, rel, ctx, lse, n_seq, L, A, nullptr, stream);
}
extern "C" int TNR_NAME(tnr_attn_long_fwd_do)(const void* qkv, const float* mask_add, const float* rel, void* ctx, float* lse,
                                              int64_t n_seq, int L, int A, const tnr_dropout_t* drop, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_attn_long_fwd")) return rc;
    TNR_CHECK_ARG(qkv && mask_add && rel && ctx && lse, "tnr_attn_long_fwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && A >= 1 && n_seq >= 1, "tnr_attn_long_fwd: need 1<=L<=512");
    const int Lr = (L + 31) / 32 * 32;
    int64_t items = n_seq * A * (Lr / 32);
    const int64_t blocks = n_seq * A * ((Lr / 32 + 3) / 4);      // one workgroup = four consecutive query tiles of one (sequence, head)
    hipLaunchKernelGGL(attn_long_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)qkv, mask_add, rel, (bf16*)ctx, lse, items, L, Lr, A, dd);
    TNR_CHECK_LAUNCH("tnr_attn_long_fwd");
    return TNR_OK;
}

extern "C" int TNR_NAME(tnr_attn_long_bwd)(const void* qkv, const float* mask_add, const float* rel, const void* ctx,
                                           const void* dctx, const float* lse, float* delta, void* dqkv, int64_t n_seq,
                                           int L, int A, void* stream) {
    return TNR_NAME(tnr_attn_long_bwd_do)(qkv, mask_add, rel, ctx, dctx, lse, delta, dqkv, n_seq, L, A, nullptr, stream);
}
extern "C" int TNR_NAME(tnr_attn_long_bwd_do)(const void* qkv, const float* mask_add, const float* rel, const void* ctx,
                                              const void* dctx, const float* lse, float* delta, void* dqkv, int64_t n_seq,
                                              int L, int A, const tnr_dropout_t* drop, void* stream) {
    TnrDrop dd;
    if (int rc = tnr_make_drop(drop, &dd, "tnr_attn_long_bwd")) return rc;
    TNR_CHECK_ARG(qkv && mask_add && rel && ctx && dctx && lse && delta && dqkv, "tnr_attn_long_bwd: null pointer");
    TNR_CHECK_ARG(L >= 1 && L <= 512 && A >= 1 && n_seq >= 1, "tnr_attn_long_bwd: need 1<=L<=512");
    const int Lr = (L + 31) / 32 * 32;
    int64_t items = n_seq * A * (Lr / 32);
    dim3 grid((unsigned)(n_seq * A * ((Lr / 32 + 3) / 4))), blk(256);      // one workgroup = four consecutive tiles of one (sequence, head)
    hipLaunchKernelGGL(attn_long_bwd_dq_kernel, grid, blk, 0, (hipStream_t)stream, (const bf16*)qkv, mask_add, rel,
                       (const bf16*)ctx, (const bf16*)dctx, lse, delta, (bf16*)dqkv, items, L, Lr, A, dd);
    TNR_CHECK_LAUNCH("tnr_attn_long_bwd/dq");
    hipLaunchKernelGGL(attn_long_bwd_dkv_kernel, grid, blk, 0, (hipStream_t)stream, (const bf16*)qkv, mask_add, rel,
                       (const bf16*)dctx, lse, delta, (bf16*)dqkv, items, L, Lr, A, dd);
    TNR_CHECK_LAUNCH("tnr_attn_long_bwd/dkv");
    return TNR_OK;
}
