/* libtnr_hip.so -- C ABI of the MI355X (gfx950) hot path of Tiny-NewsRec training.
 *
 * The reference has no FFI layer: the path sits behind Python modules and every
 * device op is a stock ATen kernel.  Each entry point
 * below replaces the ATen ops the cited reference lines launch.  Citations are
 * /root/reference/Tiny-NewsRec/<file>:<line>.
 *
 * Conventions:
 *   - plain pointers + explicit sizes; every tensor pointer is DEVICE memory
 *     owned by the caller (PyTorch allocator), including workspaces;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, a negative TNR_E* code otherwise; tnr_last_error()
 *     returns a thread-local message; no C++ exception crosses the boundary;
 *   - the library never calls hipSetDevice / hipDeviceSynchronize / hipStreamSynchronize: the caller's current device
 *     rules and nothing ever blocks the host;
 *   - device state kept by the library: ONE thing, the tile-queue counters of the persistent GEMM (see
 *     tnr_gemm_queue_reset below); everything else lives in caller-owned buffers;
 *   - 16-bit activations are bf16 (TNR_BF16); "ld" arguments are row strides in
 *     ELEMENTS; row-major everywhere.
 */
#ifndef TNR_HIP_H
#define TNR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNR_OK 0
#define TNR_EINVAL (-1)      /* bad shape / alignment / null pointer */
#define TNR_EUNSUPPORTED (-2)
#define TNR_ELAUNCH (-3)     /* hipGetLastError() after a launch */

#define TNR_BF16 0
#define TNR_F16 1            /* IEEE half: the *_f16 entry points */
#define TNR_F32 2

/* gemm epilogue flags */
#define TNR_EPI_BIAS 1       /* + bias[n] (fp32) */
#define TNR_EPI_GELU 2       /* erf-GELU (transformers BertIntermediate, call site tnlrv3/modeling.py:305) */
#define TNR_EPI_TANH 4       /* model_bert.py:25 */
#define TNR_EPI_RES 8        /* + res[m,n] (bf16) : BertSelfOutput/BertOutput residual, tnlrv3/modeling.py:287,306 */
#define TNR_EPI_MULDGELU 16  /* * gelu'(aux[m,n])  : backward of TNR_EPI_GELU */
#define TNR_EPI_OUTF32 32    /* C is fp32 instead of bf16 */
#define TNR_EPI_AUXOUT 64    /* also store the pre-activation (acc+bias) to aux (bf16) */
#define TNR_EPI_COLSUM 128   /* tnr_gemm_nt_ex: also emit per-64-row-strip column sums of the bf16 output (bias grads) */
#define TNR_EPI_DROPOUT 256  /* internal: set by tnr_gemm_nt_do when a dropout site is passed (never by the caller) */

/* Dropout site for the *_do entry points (train-mode semantics of the stage-0 / stage-1 notebooks, which call .train():
 * Post-train_KD.ipynb cell 19, Domian-specific_Post-train.ipynb cell 16; sites tnlrv3/modeling.py:177, 224 and
 * BertSelfOutput / BertOutput at :287, :306).  A host struct, read at launch; NULL or p <= 0 = no dropout (bit-identical to
 * the entry point without the suffix).  Masks are counter-based (Philox4x32-10, csrc/dropout.h): nothing is stored, the
 * backward entry points regenerate them from the same (seed, site, call). */
typedef struct tnr_dropout {
    uint64_t seed;   /* run seed = Philox key */
    uint32_t site;   /* kind | layer << 8 ; kinds: 0 embeddings, 1 attention probabilities, 2 attention-output dense, 3 FFN-output dense */
    uint32_t call;   /* number of the forward pass (distinct per encoder pass and step) */
    double p;        /* drop probability ; kept elements are scaled by 1 / (1 - p) */
} tnr_dropout_t;

int tnr_version(void);
const char* tnr_last_error(void);

/* Collectives are NOT part of this ABI: the gradient average of the reference (hvd.DistributedOptimizer / hvd.allreduce,
 * Tiny-NewsRec/run.py:141-149, utils.py:43-60) is done by the caller with torch.distributed (backend "nccl" = RCCL) on the
 * flat gradient buffer (tiny-newsrec_amd/dist.py); every entry point here is a single-device operation on the stream passed
 * last and the library holds no communicator; the only per-(device, stream) state it keeps is the GEMM's tile-queue counter
 * table described at tnr_gemm_queue_reset. */

/* ---- encoder --------------------------------------------------------------------------------- */

/* relative_position_bucket + one_hot + Linear(32->A) hoisted to one (A,32,32) fp32 table
 * (tnlrv3/modeling.py:345-373, 458-463).  weight (A,32) fp32.  Entries with i>=L or j>=L are 0.
 * In general the table is (A, Lr, Lr) with Lr = roundup(L, 32), L <= 512. */
int tnr_relpos_table(const float* weight, int A, int L, float* table, void* stream);

/* BertEmbeddings.forward (tnlrv3/modeling.py:153-178) + extended mask (:446-454), reading the padded
 * token batch directly: tok = (N, 2L) int64 rows [ids | mask] (model_bert.py:124-127).
 * out (N*L, H) bf16 ; mask_add (N,Lr) fp32 = (1-mask)*-10000, and -inf-like (-1e30) for j>=L ; Lr = roundup(L,32). */
int tnr_embed_ln_fwd(const int64_t* tok, int64_t n_seq, int L, int H, const float* word, const float* pos,
                     const float* type0, const float* gamma, const float* beta, float eps,
                     void* out, float* mask_add, void* stream);

/* Same, with the token rows gathered on the device: news_combined (n_news+1, 2L) int32 is the resident
 * table of preprocess.py:48-66 / run.py:53 and nidx (N) int32 the news indices of dataloader.py:129-138,
 * so only indices cross PCIe per step (the reference ships the gathered int64 rows, dataloader.py:152-154). */
int tnr_embed_ln_fwd_indexed(const int32_t* news_combined, const int32_t* nidx, int64_t n_seq, int L, int H,
                             const float* word, const float* pos, const float* type0, const float* gamma,
                             const float* beta, float eps, void* out, float* mask_add, void* stream);

/* C[M,N] = epilogue(A[M,K] . B[N,K]^T).  bf16 operands, fp32 MFMA accumulation.
 * Forward Linear (tnlrv3/modeling.py:236-248, transformers BertSelfOutput/BertIntermediate/BertOutput),
 * and its dgrad when B is the transposed weight copy.  N % 128 == 0, K % 64 == 0, any M >= 1.
 * Streams and threads: every entry point enqueues on `stream` and returns; calls may come from any thread.  The
 * persistent 256-column kernel hands its tiles out from a small counter block the library keeps per (device, stream) and
 * that the last workgroup of a launch zeroes again - launches of one stream are ordered, so each finds it zeroed; launches
 * on different streams use different blocks and may overlap (e.g. a collective on another stream: the workgroups that do
 * get a CU pull the tiles of those that do not). */
int tnr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                void* aux, int64_t ldaux, int flags, void* stream);

/* Same with an extra output for TNR_EPI_COLSUM: colsum_part (tnr_gemm_colsum_rows(M), N) fp32 whose rows sum
 * (tnr_reduce_rows) to the column sums of C -- the bias gradient of the Linear whose dgrad this GEMM is. */
int tnr_gemm_nt_ex(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                   void* aux, int64_t ldaux, int flags, float* colsum_part, void* stream);
int64_t tnr_gemm_colsum_rows(int64_t M);

/* Debug / test hooks.  tnr_gemm_nt_route: which kernel tnr_gemm_nt(_ex) launches for a shape on the current
 * device (the decision depends on (M, N, K, flags) and the CU count only) -- the parity tests assert it so that
 * every tile variant is pinned at the shapes the training step issues.  tnr_gemm_set_option: process-wide A/B
 * switches for tools/ ("ver", "gm", "fine_pct", "allow_fine", "bm", "nt", "pp", "tnpp", "mix"; "cus" = n: plan and size the
 * persistent GEMM grids for n compute units instead of the device's, 0 = the device's - same results, pinned by
 * test_gemm_grids_sized_for_fewer_cus_bit_exact); the library never reads the environment and the defaults are the shipped
 * configuration. */
#define TNR_ROUTE_128x128 128    /* 128x128 tile, 4 waves, 2 workgroups per CU */
#define TNR_ROUTE_256x128 2128   /* 256x128 tile (N % 256 != 0) */
#define TNR_ROUTE_256x256 256    /* 256x256 tile, 8 waves */
#define TNR_ROUTE_224x256 224    /* 224-row variant of the same kernel (fewer wasted rows per round) */
int tnr_gemm_nt_route(int64_t M, int64_t N, int64_t K, int flags);
int tnr_gemm_set_option(const char* key, int value);
/* The row tiling of the persistent 256-column kernel on a device with n_cu compute units (host arithmetic only, no device
 * needed): *panels row panels, *tall of them 32 * *mi rows high and the others 32 rows shorter, spread evenly - panel p starts
 * at row (32 * mi - 32) * p + 32 * floor(p * tall / panels).  Chosen so that panels * N / 256 tiles fill whole rounds of the
 * workgroups (option "mix" = 0: one height).  Results do not depend on the tiling (a row's K order is the same). */
int tnr_gemm_nt_plan(int64_t M, int64_t N, int flags, int n_cu, int* mi, int* panels, int* tall);
/* The persistent 256-column GEMM kernel (tnr_gemm_nt* on the TNR_ROUTE_256x256 / 224x256 routes) hands out its tiles through
 * nine 32-bit counters per (device, stream) pair, kept in a 128-entry table inside the library (a __device__ array; 288 KB).
 * A pair is bound at its first launch (its counters are zeroed by a hipMemsetAsync on that stream) and every completed launch
 * returns them to zero, so launches need no workspace argument.  The table is never drained and the current device is never
 * changed: the 129th distinct pair is refused with TNR_EUNSUPPORTED (run fewer streams, or tnr_gemm_set_option("pp", 0) for
 * the non-persistent kernel).  tnr_gemm_queue_reset zeroes the calling stream's counters again (stream-ordered): only needed
 * after a launch on that stream was aborted (a device fault survived by the process), never in normal operation.
 * (Test hooks such as the CU hog live in libtnr_testhooks.so, built beside this library for tests/ and tools/ only.) */
int tnr_gemm_queue_reset(void* stream);
/* Measurement hook (bench.py's `box` object; no reference counterpart - the reference never reports a clock): while `buf` is
 * non-NULL, workgroup b < n_pairs of every persistent NT launch (TNR_ROUTE_256x256 / 224x256) writes, as it leaves, the shader
 * cycles (s_memtime) and the 100 MHz ticks (s_memrealtime) of its life to buf[2 b], buf[2 b + 1] (uint64, device memory owned
 * by the caller, 8-byte aligned): cycles / ticks x 100 = the MHz the chip held UNDER that launch.  Process-wide like the
 * options; NULL or n_pairs = 0 turns it off (the default; two scalar loads per workgroup when on).  Results unchanged. */
int tnr_gemm_clock_stamps(void* buf, int64_t n_pairs);

/* dW[N,K] (fp32) = dY[M,N]^T . X[M,K] : weight gradient of a Linear.  Reduction over M is split into
 * `splits` slabs in `ws` (fp32, splits*N*K elements) and summed in fixed order (deterministic).
 * Rows [M, Mpad) of dY and X must be zero, Mpad = roundup(M, 64) ; N % 128 == 0, K % 128 == 0.
 * accumulate != 0 adds into dW.  _ex: dW (+)= out_scale * dY^T X (the fp16 build's backward runs on loss-scaled
 * gradients; 1 / scale is applied here so that parameter gradients are the true ones, run.py:194). */
int tnr_gemm_tn_wgrad(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                      int64_t M, int64_t N, int64_t K, float* ws, int splits, int accumulate, void* stream);
int tnr_gemm_tn_wgrad_ex(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                         int64_t M, int64_t N, int64_t K, float* ws, int splits, int accumulate, float out_scale,
                         void* stream);
int64_t tnr_gemm_tn_ws_elems(int64_t N, int64_t K, int splits);
/* Up to four weight gradients - the Linears of one encoder layer that share a backward step (modeling.py:287, 305-306 and the
 * q / k / v + output Linears of :205-231) - in ONE persistent launch + one slab-sum launch: the (split, tile) units of all of
 * them are pulled from one queue, each computed exactly as in its own tnr_gemm_tn_wgrad_ex call (same unit partition, same slab
 * order: bit-identical results).  Every problem has its own `ws`.  Problems off the 256 x 256 route (N or K not a multiple of
 * 256) make the call fall back to one launch per problem.
 * accumulate == 2 CHAINS a problem to its predecessor: another row range of the same gradient (same dW, N, K, lddw, out_scale;
 * operands and M of its own) - the two encoder passes of a stage-1 step, Post-train_KD.ipynb cell 12, whose autograd sums their
 * contributions to every shared weight.  Its `ws` is ignored: its slabs follow the predecessor's in the chain HEAD's workspace,
 * which must hold (sum of the chain's `splits`) * N * K floats, and ONE fixed-order slab sum covers the chain (the head's
 * `accumulate` decides whether dW is overwritten or added to). */
typedef struct {
    const void* dY; int64_t lddy;
    const void* X; int64_t ldx;
    float* dW; int64_t lddw;
    int64_t M, N, K;
    float* ws;               /* splits * N * K fp32, this problem's own (chained problems: see above) */
    int splits, accumulate;
    float out_scale;
} tnr_wgrad_problem_t;
int tnr_gemm_tn_wgrad_group(const tnr_wgrad_problem_t* problems, int n, void* stream);

/* LayerNorm over the last dim (H % 256 == 0), eps inside the sqrt (torch.nn.LayerNorm).
 * fwd: y = LN(x) ; stats (M,2) fp32 = (mean, rstd) kept for backward. */
int tnr_ln_fwd(const void* x, const float* gamma, const float* beta, float eps, void* y, float* stats,
               int64_t M, int H, void* stream);
/* bwd: dx = LN'(dy) ; partial sums go to part (nblk,3,H) fp32 then are reduced into dgamma, dbeta and
 * dxsum = column sums of dx (the bias gradient of the Linear in front of this LayerNorm); each may be null.
 * With all three null and part given, only the partials (nblk = ceil(M/128) rows of [dgamma|dbeta|dxsum]) are
 * written and the caller reduces them (tnr_reduce_multi). */
int tnr_ln_bwd(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
               float* dgamma, float* dbeta, float* dxsum, float* part, int64_t M, int H, void* stream);
int64_t tnr_ln_bwd_part_elems(int64_t M, int H);   /* workspace good for every row count <= M */
int64_t tnr_ln_bwd_blocks(int64_t M);              /* partial rows (nblk) written for exactly M rows */

/* BertSelfAttention.multi_head_attention (tnlrv3/modeling.py:205-231) for L <= 32, head size 64:
 * softmax(Q K^T / 8 + mask_add + rel) V, heads merged.  qkv (N*L, 3*A*64) bf16 = [q | k | v];
 * rel (A,32,32) fp32 from tnr_relpos_table ; ctx (N*L, A*64) bf16. */
int tnr_attn_l32_fwd(const void* qkv, const float* mask_add, const float* rel, void* ctx,
                     int64_t n_seq, int L, int A, void* stream);
/* backward: recomputes the probabilities ; dqkv (N*L, 3*A*64) bf16 ; bias_part (N, 3*A*64) fp32 (nullable) =
 * per-sequence column sums of dqkv (rows sum to the q/k/v bias gradient). */
int tnr_attn_l32_bwd(const void* qkv, const float* mask_add, const float* rel, const void* dctx,
                     void* dqkv, float* bias_part, int64_t n_seq, int L, int A, void* stream);

/* The same attention for longer sequences (L <= 512: stage-1 title/body matching, Post-train_KD.ipynb cell 4-14):
 * flash-style 32-key tiles with an online softmax.  mask_add (N, Lr), rel (A, Lr, Lr) from tnr_embed_ln_fwd /
 * tnr_relpos_table with Lr = roundup(L, 32); lse (N, A, Lr) fp32 out (kept for backward). */
int tnr_attn_long_fwd(const void* qkv, const float* mask_add, const float* rel, void* ctx, float* lse,
                      int64_t n_seq, int L, int A, void* stream);
/* backward in two deterministic passes (dQ per query tile, dK/dV per key tile); delta (N, A, Lr) fp32 workspace */
int tnr_attn_long_bwd(const void* qkv, const float* mask_add, const float* rel, const void* ctx, const void* dctx,
                      const float* lse, float* delta, void* dqkv, int64_t n_seq, int L, int A, void* stream);

/* column sums (bias gradients): out[n] (+)= sum_m X[m,n] ; X bf16 or fp32 (dtype) ; part (nblk,N) fp32 */
int tnr_colsum(const void* X, int64_t ldx, int dtype, int64_t M, int64_t N, float* out, float* part,
               int accumulate, void* stream);
/* batch of matrices X + z*sX -> out (batch, N) ; part needs batch * tnr_colsum_part_elems(M, N) */
int tnr_colsum_batched(const void* X, int64_t ldx, int64_t sX, int dtype, int64_t M, int64_t N, int batch,
                       float* out, float* part, int accumulate, void* stream);
int64_t tnr_colsum_part_elems(int64_t M, int64_t N);

/* ---- heads ----------------------------------------------------------------------------------- */

/* AttentionPooling over the L tokens of each title, no mask (model_bert.py:15-34 called at :133).
 * e (N*L, lde) fp32 = tanh(fc1 y) from tnr_gemm_nt (padded columns must be 0) ; w2 (Q) ; b2 scalar.
 * out: nv (N,H) fp32, alpha (N,Lr) fp32 normalised weights (Lr = roundup(L,32), L <= 512), den (N) fp32 = sum exp + 1e-8. */
int tnr_attpool_fwd(const void* y, const float* e, int64_t lde, const float* w2, const float* b2, int Q,
                    float* nv, float* alpha, float* den, int64_t n_seq, int L, int H, void* stream);
/* backward: dnv (N,H) fp32 -> dy_direct (N*L,H) bf16 = alpha*dnv ; dpre (N*L, lddpre) bf16 =
 * d tanh-preactivation (padded cols 0) ; dw2_part (N,Q) ; db2_part (N) ; db1_part (N, lddpre) nullable = per-title
 * column sums of dpre (fc1 bias gradient partials). */
int tnr_attpool_bwd(const void* y, const float* e, int64_t lde, const float* w2, int Q, const float* dnv,
                    const float* alpha, const float* den, void* dy_direct, void* dpre, int64_t lddpre,
                    float* dw2_part, float* db2_part, float* db1_part, int64_t n_seq, int L, int H, void* stream);

/* The same pooling (model_bert.py:15-34) for FEW, LONG sequences - the bodies of stage 1 (Post-train_KD.ipynb cell 12: 32 x 512
 * tokens): token-parallel parts run per (sequence, 64-token chunk) or one wave per token instead of one workgroup per sequence;
 * same arguments and outputs plus a caller-owned workspace of tnr_attpool_long_ws_elems(n_seq, L, H, Q, lddpre) floats.  Outputs
 * equal tnr_attpool_fwd / _bwd up to fp32 rounding (chunk partials are combined in chunk order: deterministic). */
int64_t tnr_attpool_long_ws_elems(int64_t n_seq, int L, int H, int Q, int64_t lddpre);
int tnr_attpool_fwd_long(const void* y, const float* e, int64_t lde, const float* w2, const float* b2, int Q,
                         float* nv, float* alpha, float* den, float* ws, int64_t n_seq, int L, int H, void* stream);
int tnr_attpool_bwd_long(const void* y, const float* e, int64_t lde, const float* w2, int Q, const float* dnv,
                         const float* alpha, void* dy_direct, void* dpre, int64_t lddpre, float* dw2_part,
                         float* db2_part, float* db1_part, float* ws, int64_t n_seq, int L, int H, void* stream);

/* NewsEncoder pooling 'cls' (mean = 0: hidden state of token 0) or mean over all L positions (mean = 1), model_bert.py:
 * 130-135: y (n_seq*L, H) 16-bit -> nv (n_seq, H) fp32 ; backward dnv -> dy (every row written). */
int tnr_pool_fwd(const void* y, float* nv, int64_t n_seq, int L, int H, int mean, void* stream);
int tnr_pool_bwd(const float* dnv, void* dy, int64_t n_seq, int L, int H, int mean, void* stream);

/* small fp32 GEMM on the f32 MFMA (exact fp32):  for z in [0,batch):
 *   C_z[m,n] = alpha * sum_k A_z(m,k) B_z(n,k) + bias_z[n] + beta * C_z[m,n]
 * A_z(m,k) = A[z*sA + m*a_rs + k*a_cs] (a_idx must be NULL), likewise B.  ksplit > 1 splits K over workgroups
 * into part (ksplit, batch, M, N) fp32 (caller's workspace) and a second kernel sums the slices in fixed order and
 * applies alpha / bias / beta: the fp32 MFMA issues one 32x32x2 block per 64 cycles, so a workgroup's time is
 * K * 32 cycles whatever the tile count -- few-tile GEMMs with a long K are latency-sized unless K is split. */
int tnr_sgemm(const float* A, int64_t a_rs, int64_t a_cs, int64_t sA, const int32_t* a_idx,
              const float* B, int64_t b_rs, int64_t b_cs, int64_t sB,
              float* C, int64_t ldc, int64_t sC, const float* bias, int64_t sBias,
              int64_t M, int64_t N, int64_t K, int batch, float alpha, float beta,
              int ksplit, float* part, void* stream);

/* Up to 8 independent tnr_sgemm problems in ONE launch (+ one launch that reduces every split problem): the heads' GEMMs are
 * latency-sized, so problems that do not depend on each other cost the longest of them instead of their sum.  Each problem is
 * computed exactly as tnr_sgemm computes it (same tiles, same K split rule, same summation order): identical bits.  The
 * problem array is HOST memory, read at the call. */
typedef struct tnr_sgemm_problem {
    const float* A; int64_t a_rs, a_cs, sA;
    const float* B; int64_t b_rs, b_cs, sB;
    float* C; int64_t ldc, sC;
    const float* bias; int64_t sBias;
    int64_t M, N, K;
    int batch;
    float alpha, beta;
    int ksplit;
    float* part;             /* ksplit > 1: (ksplit, batch, M, N) fp32 workspace of THIS problem (not shared inside a group) */
} tnr_sgemm_problem_t;
int tnr_sgemm_group(const tnr_sgemm_problem_t* problems, int n, void* stream);

/* out = [a (na) | b (nb)]: the step's news indices as one array - history slots (B*U) then candidate slots (B*C), the row
 * order of the encoder pass (dataloader.py:129-138 at index level; replaces a torch.cat inside the step) */
int tnr_concat_i32(const int32_t* a, int64_t na, const int32_t* b, int64_t nb, int32_t* out, void* stream);

/* out[z, out_row0 + r, :] = tbl[z, idx[r], :]  (dataloader.py:140-144 teacher-embedding gather, done on
 * device from resident tables instead of on the host) */
int tnr_gather_rows(const float* tbl, int64_t R, const int32_t* idx, int64_t n_idx, int D, int n_model,
                    float* out, int64_t out_rows, int64_t out_row0, void* stream);

/* In-batch de-duplication of news (SURVEY.md appendix iii: identical input rows encode to the same vector): the
 * encoder runs once per distinct news id of the step, tnr_gather_rows expands the vectors to the (B*U + B*C) slots
 * (dataloader.py:131,138 at index level) and this entry point folds the slot gradients back:
 *   out[u,:] = sum_{j in [seg[u], seg[u+1])} src[order[j],:]      (fixed order, no atomics)
 * order (n_slots) = slot ids grouped by distinct id, seg (n_seg+1) offsets; empty segments give zero rows. */
int tnr_segment_sum_rows(const float* src, const int32_t* order, const int32_t* seg, int64_t n_seg, int D,
                         float* out, void* stream);

/* UserEncoder.forward (model_bert.py:155-176, model != NRMS) + scorer bmm (:204 / :286-287) for
 * `n_model` encoders at once (student and/or frozen teachers), one workgroup per (impression, model).
 * vec: (n_model, R, D) fp32 row tables ; hidx (B,U) / cidx (B,C) int32 row ids ; mask (B,U) fp32.
 * params are stacked per model: pad (n_model,D), w1 (n_model,Q,D), b1 (n_model,Q), w2 (n_model,Q), b2 (n_model).
 * epre (n_model, B*U, Q) = fc1 pre-activations v W1^T + b1 of every history slot in position order (one batched
 * tnr_sgemm) ; epad (n_model, Q) = fc1(pad_doc) of every model, or NULL to have each workgroup compute it ;
 * epre == NULL: fc1 runs INSIDE the kernel on the fp32 MFMA (one launch per pass; needs D % 8 == 0 and
 * 64 (D + 4) + U Q + D + 64 floats of LDS), epad is ignored ;
 * out: user (model z at user + z*user_stride, (B,D)), score (n_model,B,C), saved e (n_model,B,U,Q), alpha (n_model,B,U), den (n_model,B). */
int tnr_user_score_fwd(const float* vec, int64_t R, const int32_t* hidx, const int32_t* cidx, const float* mask,
                       const float* pad, const float* w1, const float* b1, const float* w2, const float* b2,
                       int user_log_mask, const float* epre, const float* epad, float* user, int64_t user_stride, float* score,
                       float* e, float* alpha, float* den, int n_model, int B, int U, int C, int D, int Q, void* stream);
/* backward of the student's user encoder (model_bert.py:155-176), in two kernels around two fp32 GEMMs the caller
 * issues with tnr_sgemm:
 *   tnr_user_bwd_pre : hv (B*U, D) = blended history rows in position order ; dpre (B*U, Q) = gradient of the fc1
 *                      pre-activation ; part (B, 2Q + D + 1) = per-impression partials [b1 | w2 | pad | b2]
 *   caller           : dW1 (Q,D) = dpre^T hv ;  dhv (B*U, D) = dpre W1
 *   tnr_user_bwd_post: dvec[hidx[b,u]] += (alpha * duser + dhv) * m ; fills the pad_doc partial of `part`. */
int tnr_user_bwd_pre(const float* vec, const int32_t* hidx, const float* mask, const float* pad, const float* w2,
                     int user_log_mask, const float* duser, const float* e, const float* alpha, float* hv,
                     float* dpre, float* part, int B, int U, int D, int Q, void* stream);
int tnr_user_bwd_post(const float* dhv, const float* alpha, const float* duser, const float* mask,
                      const int32_t* hidx, int user_log_mask, float* dvec, float* part, int B, int U, int D, int Q,
                      void* stream);
int64_t tnr_user_bwd_part_stride(int D, int Q);
/* NRMS user encoder (args.model == 'NRMS': model_bert.py:37-100, 145-148, 162-164, 171-173), d_k = d_v = 16, for
 * `n_model` encoders at once (student and/or frozen teachers); fp32.
 *   tnr_user_blend_fwd : hv (n_model, B*U, D) = vec[hidx] * m + pad * (1-m) (user_log_mask 0) | vec[hidx] (1)
 *   caller             : qkv (n_model, B*U, 3*Dh) = hv [W_Q;W_K;W_V]^T + b   (tnr_sgemm), Dh = n_heads*16
 *   tnr_nrms_attn_fwd  : ctx (n_model, B*U, Dh) ; sc = exp(q.k/4) [* mask_j if use_mask] / (sum + 1e-8), raw exp (:51-58)
 *   tnr_nrms_attn_bwd  : dctx (B*U, Dh) -> dqkv (B*U, 3*Dh) for one model, recomputing sc (two fixed-order phases)
 *   tnr_user_blend_bwd : dvec[hidx] += dhv * m ; pad_part[b*part_stride + d] = sum_u dhv * (1-m) */
int tnr_user_blend_fwd(const float* vec, int64_t R, const int32_t* hidx, const float* mask, const float* pad,
                       int user_log_mask, float* hv, int n_model, int B, int U, int D, void* stream);
int tnr_user_blend_bwd(const float* dhv, const float* mask, const int32_t* hidx, int user_log_mask, float* dvec,
                       float* pad_part, int64_t part_stride, int B, int U, int D, void* stream);
int tnr_nrms_attn_fwd(const float* qkv, const float* mask, int use_mask, float* ctx, int64_t ctx_rows, int n_model,
                      int B, int U, int n_heads, void* stream);   /* ctx rows of model z start at z*ctx_rows (>= B*U) */
int tnr_nrms_attn_bwd(const float* qkv, const float* mask, int use_mask, const float* dctx, float* dqkv, int B,
                      int U, int n_heads, void* stream);

/* backward of the scorer bmm (model_bert.py:204): dvec[cidx[b,c]] += dscore[b,c]*user[b] ;
 * duser[b] += sum_c dscore[b,c]*vec[cidx[b,c]] */
int tnr_score_bwd(const float* vec, const int32_t* cidx, const float* user, const float* dscore, float* dvec,
                  float* duser, int B, int C, int D, void* stream);

/* Model.forward losses (model_bert.py:271, 288-305; kd_ce_loss :208-219): teacher CE -> weights
 * softmax(-CE) -> mixed soft labels -> distill + coef*target, and d/d student_score.
 * s_score (B,C) ; t_score (T,B,C) ; out: tw (B,T), dscore (B,C), losses[0..1] = distill, target. */
int tnr_kd_score_loss(const float* s_score, const float* t_score, const int64_t* label, float temperature,
                      float coef, float* tw, float* dscore, float* losses, int B, int C, int T, void* stream);
/* embedding KD (model_bert.py:277-284, 300-303) on stacked rows [B*U history | B*C candidate | B user]:
 * S (Rtot,D) student rows ; P (T,Rtot,D) projected teacher rows ; tw (B,T) ; Rtot = B*(U+C+1).
 * U = 0 is the stage-1 layout [B*C titles | B bodies] (Post-train_KD.ipynb cell 14).
 * out: emb loss (scalar) ; dS (Rtot,D) ; dP (T,Rtot,D) ; part (Rtot) workspace. */
int tnr_kd_embed_loss(const float* S, const float* P, const float* tw, float* loss, float* dS, float* dP,
                      float* part, int B, int U, int C, int D, int T, void* stream);

/* out[i] (+)= sum_r part[r*stride + i] , i < n  (fixed order).  `part` is scratch: tall inputs are first summed
 * in place per row chunk, so its contents are clobbered. */
int tnr_reduce_rows(const float* part, int64_t rows, int64_t stride, int64_t n, float* out, int accumulate,
                    void* stream);

/* many fixed-order row reductions in ONE launch: desc = n_blocks x 6 int64 on the device, one per workgroup,
 * {src ptr, rows, row stride in floats, ncols <= 64, dst ptr, accumulate | (fp32 bits of scale) << 32}:
 * dst[c] (+)= scale * sum_r src[r*stride + c] ; upper half 0 = scale 1.
 * Callers reduce tall partial matrices in two launches (row chunks in place, then the chunk rows). */
int tnr_reduce_multi(const int64_t* desc, int n_blocks, void* stream);
/* x[i] *= s , i < n (fp32) */
int tnr_scale_inplace(float* x, int64_t n, float s, void* stream);

/* ---- dropout (train-mode) variants: the same operations with a dropout site, see tnr_dropout_t ------------------
 * The two embedding variants (mask after the LayerNorm, tnlrv3/modeling.py:177) also take pos_ids: NULL = position i of token i
 * (BERT / UniLM), else an int32 table laid out like the token table (n_seq or n+1 rows x L) naming each token's position row --
 * RoBERTa's cumulative non-pad count + padding_idx (PLM-NR --model_type roberta, PLM-NR/utils.py:17-21). */
int tnr_embed_ln_fwd_do(const int64_t* tok, int64_t n_seq, int L, int H, const float* word, const float* pos,
                        const float* type0, const float* gamma, const float* beta, float eps, void* out, float* mask_add,
                        const tnr_dropout_t* drop, const int32_t* pos_ids, void* stream);
int tnr_embed_ln_fwd_indexed_do(const int32_t* news_combined, const int32_t* nidx, int64_t n_seq, int L, int H,
                                const float* word, const float* pos, const float* type0, const float* gamma,
                                const float* beta, float eps, void* out, float* mask_add, const tnr_dropout_t* drop,
                                const int32_t* pos_ids, void* stream);
/* C = epilogue(A . B^T) with the site's mask on (acc + bias [-> activation]) BEFORE the residual add:
 * BertSelfOutput / BertOutput = dense -> dropout -> LayerNorm(x + residual).  Mask element index = m * N + n. */
int tnr_gemm_nt_do(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                   void* aux, int64_t ldaux, int flags, float* colsum_part, const tnr_dropout_t* drop, void* stream);
/* LayerNorm backward behind such a Linear: dx (the residual branch's gradient) and dxm = dx * mask / (1 - p) (the Linear's
 * output gradient: what its weight gradient, dgrad and bias gradient consume; the dxsum partials are sums of dxm). */
int tnr_ln_bwd_do(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                  float* dgamma, float* dbeta, float* dxsum, float* part, int64_t M, int H, void* dxm,
                  const tnr_dropout_t* drop, void* stream);
/* attention with dropout on the normalised probabilities (:224); backward regenerates the mask */
int tnr_attn_l32_fwd_do(const void* qkv, const float* mask_add, const float* rel, void* ctx, int64_t n_seq, int L, int A,
                        const tnr_dropout_t* drop, void* stream);
int tnr_attn_l32_bwd_do(const void* qkv, const float* mask_add, const float* rel, const void* dctx, void* dqkv,
                        float* bias_part, int64_t n_seq, int L, int A, const tnr_dropout_t* drop, void* stream);
int tnr_attn_long_fwd_do(const void* qkv, const float* mask_add, const float* rel, void* ctx, float* lse, int64_t n_seq,
                         int L, int A, const tnr_dropout_t* drop, void* stream);
int tnr_attn_long_bwd_do(const void* qkv, const float* mask_add, const float* rel, const void* ctx, const void* dctx,
                         const float* lse, float* delta, void* dqkv, int64_t n_seq, int L, int A,
                         const tnr_dropout_t* drop, void* stream);
/* the multipliers (0 or 1 / (1 - p)) of a site as fp32, for tests: row-major (rows, cols) sites, and the attention
 * probabilities (pairs = n_seq * A, L, L) through the per-query (by_columns = 0) or per-key (1) device accessor */
int tnr_dropout_mask(const tnr_dropout_t* drop, int64_t rows, int64_t cols, float* out, void* stream);
int tnr_dropout_mask_probs(const tnr_dropout_t* drop, int64_t pairs, int L, int by_columns, float* out, void* stream);

/* ---- optimiser ------------------------------------------------------------------------------- */

/* torch.optim.Adam(amsgrad=True) (run.py:134) on a flat fp32 parameter buffer ; step = 1-based count.
 * grad_scale multiplies g first (1/world for the data-parallel average, run.py:145-149).
 * vmax == NULL: plain Adam (Post-train_KD.ipynb cell 18). */
int tnr_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, int step,
                     float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* Dynamic loss scaling of the fp16 build (the reference trains in fp32 and needs none, run.py:134,194-195; with 16-bit
 * activation gradients one overflow would otherwise poison m / v / vmax for good).  guard = 4 uint32 words on the device,
 * zero at start: [0] stamp of the last step that overflowed, [1] number of skipped steps, [2], [3] unused.
 * tnr_grad_nonfinite: if any of g[0 .. n) is inf or nan, guard[0] = max(guard[0], stamp) and guard[1] += 1 (stamp >= 1, the
 * caller's running count of optimiser steps; two launches: the scan, in which only a workgroup that found something touches
 * guard, and a one-thread kernel that counts the skip).  tnr_amsgrad_step_guarded = tnr_amsgrad_step, except that (a) a launch whose
 * stamp equals guard[0] touches nothing: the skipped step; (b) Adam's bias corrections use step - (guard[1] - known_skips):
 * `step` is the caller's count of steps, known_skips how many skipped ones it has already taken out of that count (it learns of
 * a skip with a lag of up to two steps).  Stream-ordered, no host synchronisation; the caller reads guard back whenever it
 * likes (tiny-newsrec_amd/engine.py: LossScaler).  guard == NULL: unguarded. */
int tnr_grad_nonfinite(const float* g, int64_t n, unsigned* guard, unsigned stamp, void* stream);
/* the same in pieces, for a gradient that arrives bucket by bucket (data parallelism: each bucket is scanned as soon as its
 * all-reduce has landed, beside the collectives still in flight): any number of _scan launches over disjoint slices with one
 * stamp (they only ever raise guard[0] to it), then ONE _commit that counts the skipped step.  scan(all) + commit ==
 * tnr_grad_nonfinite. */
int tnr_grad_nonfinite_scan(const float* g, int64_t n, unsigned* guard, unsigned stamp, void* stream);
int tnr_grad_nonfinite_commit(unsigned* guard, unsigned stamp, void* stream);
int tnr_amsgrad_step_guarded(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, int step,
                             float lr, float beta1, float beta2, float eps, float grad_scale, const unsigned* guard,
                             unsigned stamp, unsigned known_skips, void* stream);

/* refresh bf16 weight copies after an update: desc = n_desc * 8 int64 on DEVICE:
 * {src fp32 ptr, rows, cols, dst ptr (or 0), dst ld, dstT ptr (or 0), dstT ld, unused}
 * dst[r*ld + c] = bf16(src[r,c]) ; dstT[c*ldT + r] = bf16(src[r,c]). */
int tnr_refresh_shadows(const int64_t* desc, int n_desc, int64_t total_tiles, const int64_t* tile_start,
                        void* stream);

/* elementwise helpers */
int tnr_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int tnr_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);

/* ---- fp16 activations ------------------------------------------------------------------------------
 * Every entry point that touches 16-bit tensors exists a second time with the suffix _f16: identical
 * signature and semantics with IEEE half instead of bf16 (same MFMA rate, 3 more mantissa bits: the build used
 * for the 1e-3 parity bound).  The sources are compiled twice (-DTNR_BUILD_F16). */
int tnr_embed_ln_fwd_f16(const int64_t* tok, int64_t n_seq, int L, int H, const float* word, const float* pos,
                     const float* type0, const float* gamma, const float* beta, float eps,
                     void* out, float* mask_add, void* stream);
int tnr_embed_ln_fwd_indexed_f16(const int32_t* news_combined, const int32_t* nidx, int64_t n_seq, int L, int H,
                             const float* word, const float* pos, const float* type0, const float* gamma,
                             const float* beta, float eps, void* out, float* mask_add, void* stream);
int tnr_gemm_nt_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                void* aux, int64_t ldaux, int flags, void* stream);
int tnr_gemm_nt_ex_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                   void* aux, int64_t ldaux, int flags, float* colsum_part, void* stream);
int64_t tnr_gemm_colsum_rows_f16(int64_t M);
int tnr_gemm_nt_route_f16(int64_t M, int64_t N, int64_t K, int flags);
int tnr_gemm_tn_wgrad_f16(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                      int64_t M, int64_t N, int64_t K, float* ws, int splits, int accumulate, void* stream);
int tnr_gemm_tn_wgrad_ex_f16(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, int64_t lddw,
                         int64_t M, int64_t N, int64_t K, float* ws, int splits, int accumulate, float out_scale,
                         void* stream);
int64_t tnr_gemm_tn_ws_elems_f16(int64_t N, int64_t K, int splits);
int tnr_gemm_tn_wgrad_group_f16(const tnr_wgrad_problem_t* problems, int n, void* stream);
int tnr_ln_fwd_f16(const void* x, const float* gamma, const float* beta, float eps, void* y, float* stats,
               int64_t M, int H, void* stream);
int tnr_ln_bwd_f16(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
               float* dgamma, float* dbeta, float* dxsum, float* part, int64_t M, int H, void* stream);
int tnr_pool_fwd_f16(const void* y, float* nv, int64_t n_seq, int L, int H, int mean, void* stream);
int tnr_pool_bwd_f16(const float* dnv, void* dy, int64_t n_seq, int L, int H, int mean, void* stream);
int tnr_attn_long_fwd_f16(const void* qkv, const float* mask_add, const float* rel, void* ctx, float* lse,
                          int64_t n_seq, int L, int A, void* stream);
int tnr_attn_long_bwd_f16(const void* qkv, const float* mask_add, const float* rel, const void* ctx, const void* dctx,
                          const float* lse, float* delta, void* dqkv, int64_t n_seq, int L, int A, void* stream);
int tnr_attn_l32_fwd_f16(const void* qkv, const float* mask_add, const float* rel, void* ctx,
                     int64_t n_seq, int L, int A, void* stream);
int tnr_attn_l32_bwd_f16(const void* qkv, const float* mask_add, const float* rel, const void* dctx,
                     void* dqkv, float* bias_part, int64_t n_seq, int L, int A, void* stream);
int tnr_colsum_f16(const void* X, int64_t ldx, int dtype, int64_t M, int64_t N, float* out, float* part,
               int accumulate, void* stream);
int tnr_colsum_batched_f16(const void* X, int64_t ldx, int64_t sX, int dtype, int64_t M, int64_t N, int batch,
                       float* out, float* part, int accumulate, void* stream);
int tnr_attpool_fwd_f16(const void* y, const float* e, int64_t lde, const float* w2, const float* b2, int Q,
                    float* nv, float* alpha, float* den, int64_t n_seq, int L, int H, void* stream);
int tnr_attpool_bwd_f16(const void* y, const float* e, int64_t lde, const float* w2, int Q, const float* dnv,
                    const float* alpha, const float* den, void* dy_direct, void* dpre, int64_t lddpre,
                    float* dw2_part, float* db2_part, float* db1_part, int64_t n_seq, int L, int H, void* stream);
int64_t tnr_attpool_long_ws_elems_f16(int64_t n_seq, int L, int H, int Q, int64_t lddpre);
int tnr_attpool_fwd_long_f16(const void* y, const float* e, int64_t lde, const float* w2, const float* b2, int Q,
                         float* nv, float* alpha, float* den, float* ws, int64_t n_seq, int L, int H, void* stream);
int tnr_attpool_bwd_long_f16(const void* y, const float* e, int64_t lde, const float* w2, int Q, const float* dnv,
                         const float* alpha, void* dy_direct, void* dpre, int64_t lddpre, float* dw2_part,
                         float* db2_part, float* db1_part, float* ws, int64_t n_seq, int L, int H, void* stream);
int tnr_refresh_shadows_f16(const int64_t* desc, int n_desc, int64_t total_tiles, const int64_t* tile_start,
                        void* stream);
int tnr_cast_f32_to_bf16_f16(const float* src, void* dst, int64_t n, void* stream);
int tnr_cast_bf16_to_f32_f16(const void* src, float* dst, int64_t n, void* stream);

int tnr_embed_ln_fwd_do_f16(const int64_t* tok, int64_t n_seq, int L, int H, const float* word, const float* pos,
                        const float* type0, const float* gamma, const float* beta, float eps, void* out, float* mask_add,
                        const tnr_dropout_t* drop, const int32_t* pos_ids, void* stream);
int tnr_embed_ln_fwd_indexed_do_f16(const int32_t* news_combined, const int32_t* nidx, int64_t n_seq, int L, int H,
                                const float* word, const float* pos, const float* type0, const float* gamma,
                                const float* beta, float eps, void* out, float* mask_add, const tnr_dropout_t* drop,
                                const int32_t* pos_ids, void* stream);
int tnr_gemm_nt_do_f16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, const float* bias, const void* res, int64_t ldres,
                   void* aux, int64_t ldaux, int flags, float* colsum_part, const tnr_dropout_t* drop, void* stream);
int tnr_ln_bwd_do_f16(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                  float* dgamma, float* dbeta, float* dxsum, float* part, int64_t M, int H, void* dxm,
                  const tnr_dropout_t* drop, void* stream);
int tnr_attn_l32_fwd_do_f16(const void* qkv, const float* mask_add, const float* rel, void* ctx, int64_t n_seq, int L, int A,
                        const tnr_dropout_t* drop, void* stream);
int tnr_attn_l32_bwd_do_f16(const void* qkv, const float* mask_add, const float* rel, const void* dctx, void* dqkv,
                        float* bias_part, int64_t n_seq, int L, int A, const tnr_dropout_t* drop, void* stream);
int tnr_attn_long_fwd_do_f16(const void* qkv, const float* mask_add, const float* rel, void* ctx, float* lse, int64_t n_seq,
                         int L, int A, const tnr_dropout_t* drop, void* stream);
int tnr_attn_long_bwd_do_f16(const void* qkv, const float* mask_add, const float* rel, const void* ctx, const void* dctx,
                         const float* lse, float* delta, void* dqkv, int64_t n_seq, int L, int A,
                         const tnr_dropout_t* drop, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Collectives of the data-parallel step.  Replace hvd.init / hvd.broadcast_parameters /
 * hvd.broadcast_optimizer_state (Tiny-NewsRec/run.py:141-144, utils.py:43-60) and the gradient all-reduce inside
 * hvd.DistributedOptimizer (run.py:145-149: average of the trainable parameters' gradients, fp32, no compression).
 * One process per GPU; the communicator lives on the caller's CURRENT device (the library never calls hipSetDevice); every
 * collective is asynchronous on `stream` and in place; RCCL is bound at run time (dlopen librccl.so.1: inside a PyTorch-ROCm
 * process the copy torch has loaded) - without it these return TNR_EUNSUPPORTED and nothing else in the library is affected.
 * tnr_comm_unique_id: rank 0 fills 128 bytes that reach the other ranks out of band (the host side uses the rendezvous store
 * torch.distributed already has, dist.py).  tnr_comm_allreduce_avg: average = 1 divides by the world size inside the reduction
 * (ncclAvg), 0 sums (the engine folds 1 / world into tnr_amsgrad_step's grad_scale).  tnr_comm_reduce_scatter_allgather: the same
 * result by direct exchange on the fully connected xGMI mesh (n a multiple of the world size, `shard` n / world floats of the
 * caller's).  Errors: RCCL's own code and text in tnr_last_error(), status TNR_ELAUNCH. */
typedef struct tnr_comm tnr_comm_t;
int tnr_comm_unique_id(void* id128);
int tnr_comm_init(const void* id128, int world, int rank, tnr_comm_t** comm);
int tnr_comm_world(const tnr_comm_t* comm, int* world, int* rank);
int tnr_comm_broadcast(tnr_comm_t* comm, float* buf, int64_t n, int root, void* stream);
int tnr_comm_allreduce_avg(tnr_comm_t* comm, float* buf, int64_t n, int average, void* stream);
int tnr_comm_reduce_scatter_allgather(tnr_comm_t* comm, float* buf, float* shard, int64_t n, int average, void* stream);
int tnr_comm_destroy(tnr_comm_t* comm);

#ifdef __cplusplus
}
#endif
#endif
