// Counter-based dropout masks (train-mode semantics of the stage-0 / stage-1 notebooks: Post-train_KD.ipynb cell 19,
// Domian-specific_Post-train.ipynb cell 16 call .train(), so hidden_dropout_prob / attention_probs_dropout_prob of the
// UniLM config are live at tnlrv3/modeling.py:177, 224 and in BertSelfOutput / BertOutput, call sites :287, :306).
// Philox4x32-10 keyed by the run's seed, counter = (call index lo, call index hi, site, forward-call number); one call
// gives eight 16-bit uniforms; an element is kept iff its uniform >= thresh = floor(p * 65536 + 0.5) and kept elements are
// scaled by 1 / (1 - p).  Nothing is stored: the backward kernels regenerate the same bits from the same counter.
//   row-major (rows, cols) sites:  element e = row * cols + col -> call e >> 3, uniform e & 7
//   attention probabilities:       4 x 4 (query, key) blocks -> call ((pair * nb + q >> 2) * nb + k >> 2) * 2 + ((q & 3) >> 1),
//                                  uniform ((q & 1) << 2) | (k & 3) ; pair = n * A + a, nb = Lr / 4
// oracle/dropout_oracle.py restates this bit for bit (tnr_dropout_mask dumps the multipliers for the tests).
#pragma once
#include <stdint.h>

struct TnrDrop {
    uint32_t k0, k1;       // seed
    uint32_t site, call;   // counter words 2, 3
    uint32_t thresh;       // 0 = dropout off
    float scale;           // 1 / (1 - p)
};

#define TNR_DROP_EMB 0
#define TNR_DROP_PROB 1
#define TNR_DROP_ATTN_OUT 2
#define TNR_DROP_FFN_OUT 3

#if defined(__HIPCC__)
__device__ __forceinline__ void tnr_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// the eight multipliers (0 or scale) of Philox call `ci`
__device__ __forceinline__ void tnr_drop8(const TnrDrop& d, uint64_t ci, float (&m)[8]) {
    uint32_t o[4];
    tnr_philox((uint32_t)ci, (uint32_t)(ci >> 32), d.site, d.call, d.k0, d.k1, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = ((o[e >> 1] >> (16 * (e & 1))) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
}

// four consecutive elements e0 .. e0 + 3 of a row-major site (e0 % 4 == 0)
__device__ __forceinline__ void tnr_drop4(const TnrDrop& d, uint64_t e0, float (&m)[4]) {
    uint32_t o[4];
    const uint64_t ci = e0 >> 3;
    tnr_philox((uint32_t)ci, (uint32_t)(ci >> 32), d.site, d.call, d.k0, d.k1, o);
    const uint32_t w0 = (e0 & 4) ? o[2] : o[0], w1 = (e0 & 4) ? o[3] : o[1];
    m[0] = (w0 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m[1] = (w0 >> 16) >= d.thresh ? d.scale : 0.f;
    m[2] = (w1 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m[3] = (w1 >> 16) >= d.thresh ? d.scale : 0.f;
}

// attention probabilities, one query x four consecutive keys (k0 % 4 == 0): multipliers of (q, k0 .. k0 + 3)
__device__ __forceinline__ void tnr_drop_prob_row(const TnrDrop& d, uint64_t pair, int nb, int q, int k0, float (&m)[4]) {
    uint32_t o[4];
    const uint64_t ci = ((pair * nb + (q >> 2)) * nb + (k0 >> 2)) * 2 + ((q & 3) >> 1);
    tnr_philox((uint32_t)ci, (uint32_t)(ci >> 32), d.site, d.call, d.k0, d.k1, o);
    const uint32_t w0 = (q & 1) ? o[2] : o[0], w1 = (q & 1) ? o[3] : o[1];
    m[0] = (w0 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m[1] = (w0 >> 16) >= d.thresh ? d.scale : 0.f;
    m[2] = (w1 & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    m[3] = (w1 >> 16) >= d.thresh ? d.scale : 0.f;
}
// ... four consecutive queries x one key (q0 % 4 == 0): multipliers of (q0 .. q0 + 3, k)
__device__ __forceinline__ void tnr_drop_prob_col(const TnrDrop& d, uint64_t pair, int nb, int q0, int k, float (&m)[4]) {
    const uint64_t c0 = ((pair * nb + (q0 >> 2)) * nb + (k >> 2)) * 2;
    const int sh = 16 * (k & 1), wi = (k & 3) >> 1;          // uniform (q & 1) * 4 + (k & 3): word 2 (q & 1) + wi, half k & 1
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
        uint32_t o[4];
        tnr_philox((uint32_t)(c0 + hq), (uint32_t)((c0 + hq) >> 32), d.site, d.call, d.k0, d.k1, o);
        const uint32_t a = wi ? o[1] : o[0], b = wi ? o[3] : o[2];
        m[2 * hq] = ((a >> sh) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
        m[2 * hq + 1] = ((b >> sh) & 0xFFFFu) >= d.thresh ? d.scale : 0.f;
    }
}
#endif
