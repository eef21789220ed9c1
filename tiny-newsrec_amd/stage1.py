"""Stage-1 knowledge distillation (title/body matching post-training) on the HIP engine.

Mirrors DistillModel.forward of the reference's Post-train_KD.ipynb (cells 12-14):
    body_vec  = news_encoder(body)                       (B, D)        bodies of up to 512 tokens
    title_vec = news_encoder(title)                      (B, 1+K, D)
    score     = bmm(title_vec, body_vec)                 (B, 1+K)
    loss      = CE(score, label) + kd_ce(mixed teacher scores, tau=1) + sum_t w_t * (MSE_title + MSE_body)
with per-sample teacher weights w = softmax(-CE(teacher score)).  The list*tensor product the notebook writes
for the weighted sum is evaluated as the stacked product it intends (oracle/newsrec_oracle.py:distill_fwd pins
that reading against the notebook's own modules).

One model, two sequence lengths: two Engine instances share parameters / gradients / optimiser state / 16-bit
weight copies and own their workspaces.  Their backwards run in step, layer by layer: the bias / LayerNorm sums of the title pass
are written and those of the body pass added; every weight gradient is ONE chained problem over the title rows and the body rows
(tnr_gemm_tn_wgrad_group, accumulate = 2).
Round 6 - JOINT passes (`Stage1Engine.joint`, the default without dropout): a Linear, a LayerNorm and a weight gradient act on token
rows one by one, so the body pass's rows are laid directly BEHIND the title pass's in every per-token buffer and each of them is
ONE launch over M = N Lt + B Lb rows instead of one per pass (8 896 rows at 30 / 128, 24 064 at 24 / 512: tile orders large enough
for the persistent 256-wide kernels); only what depends on the sequence length - embeddings, attention, pooling - still runs per
pass, on its row range (one stream: `joint_streams` puts the body's on a second one, measured slower).  A layer's four weight
gradients leave in ONE grouped persistent launch whose units share one round (`_wgrad_flush_joint`).  Every row goes through the same K order as in its own
launch: scores and losses are bit-identical to the two-launch form; parameter gradients are the same sums in another order.
Student rows live in one table S = [B*(1+K) title rows | B body rows], the layout tnr_kd_embed_loss and
tnr_score_bwd already use in stage 2 with the body vector in the "user" slot.

Trainable set = the notebook's (cell 17): heads + transform matrices + encoder layers 2 and 3 of the frozen UniLM
(`trainable_layers`); optimiser = its plain Adam with two rates (cell 18: 1e-6 for bert_model, 1e-5 for the rest) via
step(lr, lr_bert, amsgrad=False).  The notebook trains under .train() (cell 19:6), i.e. with the UniLM config's dropout
(hidden 0.1, attention probabilities 0.1) live in the encoder: set_dropout() turns the same four sites on (counter-based masks,
csrc/dropout.h; the title and the body pass draw independent masks, as two passes through nn.Dropout do).
"""
import torch

import tnr_hip as T
from engine import BERT, PFX, QPAD, Engine, EngineConfig, _ReduceBatch, layer_param_order


class Stage1Engine:
    def __init__(self, n_layers=4, trainable_layers=(0, 1, 2, 3), num_teachers=4, npratio=4, title_len=30, body_len=256,
                 device="cuda:0", batch=32, dtype="fp16", **dims):
        """dims: hidden, heads, inter, news_dim, news_query, vocab, ... (EngineConfig keywords)."""
        # num_teachers = 0 is stage 0: TitleBodySimModel of Domian-specific_Post-train.ipynb (cells 10-11), plain CE
        common = dict(n_layers=n_layers, trainable_layers=trainable_layers, num_teachers=num_teachers, user_log_length=0,
                      temperature=1.0, coef=1.0, stage1=True, **dims)
        self.cfg_t = EngineConfig(npratio=npratio, num_words=title_len, **common)
        self.cfg_b = EngineConfig(npratio=0, num_words=body_len, **common)
        # the title engine's per-token / per-sequence workspaces have room for the body pass's rows / sequences behind its own (joint passes)
        self.title = Engine(self.cfg_t, device, max_batch=batch, dtype=dtype, extra_rows=batch * body_len, extra_seqs=batch)
        self.body = Engine(self.cfg_b, device, max_batch=batch, dtype=dtype, share=self.title)
        self.dev = self.title.dev
        self.ws = None               # slabs of the chained weight gradients (allocated by the first backward)

    def set_dropout(self, p_hidden, p_attn, seed):
        """Train-mode dropout of both encoder passes (tnlrv3/config/*.json:2,4 under Post-train_KD.ipynb cell 19:6)."""
        self.title.set_dropout(p_hidden, p_attn, seed, pass_id=0)
        self.body.set_dropout(p_hidden, p_attn, seed, pass_id=1)

    # parameters / optimiser state are the title engine's
    def load_state_dict(self, sd):
        self.title.load_state_dict(sd)
        self.body.refresh_rel()

    def state_dict(self):
        return self.title.state_dict()

    def grad(self, name):
        return self.title.grad(name)

    @property
    def shapes(self):
        return self.title.shapes

    def forward_indexed(self, title_table, body_table, idx, label, t_title_tables, t_body_tables, body_idx=None):
        """The same step fed by index (DistillDataset.__getitem__, cell 8, at index level): title_table (n, 2Lt) and
        body_table (n, 2Lb) int32 token tables and teacher tables (T, n, D) fp32 stay in HBM; idx (B, 1+K) int32 holds the
        positive document first, then its sampled negatives (the body is the positive's).  body_idx (B,) int32 contiguous = idx[:, 0]
        if the loader has it (saves the step its one strided copy)."""
        t, b = self.title, self.body
        cfg = self.cfg_t
        B = idx.shape[0]
        C, D, T_ = cfg.C, cfg.D, cfg.T
        assert idx.shape[1] == C
        self._prepare(B)
        self.ran_joint = self._joint_ok()      # (before anything is copied into the workspaces: it may lay them out afresh)
        N, Rt = B * C, B * C + B
        self.cur = (B, N, Rt)
        t.label = label.to(torch.int64).contiguous()
        tidx = idx.reshape(-1).to(torch.int32).contiguous()
        bidx = idx[:, 0].to(torch.int32).contiguous() if body_idx is None else body_idx
        def teacher_side():
            # the teacher side (row gathers, teacher scores, projections) needs nothing of the student's: in front of the title pass,
            # where it runs beside the start of the body pass instead of alone between the encoders and the losses
            if T_:
                T.call("tnr_gather_rows", t_title_tables, t_title_tables.shape[1], tidx, N, D, T_, t.X, t.X.shape[1], 0)
                T.call("tnr_gather_rows", t_body_tables, t_body_tables.shape[1], bidx, B, D, T_, t.X, t.X.shape[1], N)
                self._teacher_side(B, N, Rt)

        def title_pass():
            teacher_side()
            t.encode(title_table, N, nidx=tidx)
        if self.ran_joint:
            self._encode_joint(B, N, title_table, tidx, body_table, bidx, teacher_side)
        else:
            self._encode_both(lambda: b.encode(body_table, B, nidx=bidx, out=t.S[N:]), title_pass)
        return self._heads(B, N, Rt)

    def forward(self, title, body, label, teacher_titles, teacher_bodies):
        """title (B,1+K,2Lt) / body (B,2Lb) int64 [ids | mask]; label (B,); teacher_* lists of (B,1+K,D) / (B,D) fp32
        (or stacked (T,B,1+K,D) / (T,B,D)).  -> (losses [distill, target, emb, -], score (B,1+K))."""
        t, b = self.title, self.body
        cfg = self.cfg_t
        B = title.shape[0]
        C, D, T_ = cfg.C, cfg.D, cfg.T
        assert title.shape[1:] == (C, 2 * cfg.L) and body.shape == (B, 2 * self.cfg_b.L)
        self._prepare(B)
        self.ran_joint = self._joint_ok()      # (before anything is copied into the workspaces: it may lay them out afresh)
        N, Rt = B * C, B * C + B
        self.cur = (B, N, Rt)
        t.label = label.to(torch.int64).contiguous()
        b.tok[:B].copy_(body)
        t.tok[:N].copy_(title.reshape(N, 2 * cfg.L))
        for i in range(T_):
            t.X[i, :N].copy_(teacher_titles[i].reshape(N, D))
            t.X[i, N:Rt].copy_(teacher_bodies[i].reshape(B, D))

        def teacher_side():
            if T_:
                self._teacher_side(B, N, Rt)

        def title_pass():
            teacher_side()
            t.encode(t.tok[:N], N)
        if self.ran_joint:
            self._encode_joint(B, N, t.tok[:N], None, b.tok[:B], None, teacher_side)
        else:
            self._encode_both(lambda: b.encode(b.tok[:B], B, out=t.S[N:]), title_pass)    # cell 12 encodes the bodies first
        return self._heads(B, N, Rt)

    # The body pass on a second stream beside the title pass: most launches of either pass are partial rounds (4 800 / 4 096 token
    # rows at 30 / 128: 228 / 192 tiles for 256 CUs), side by side they fill the chip.  The same kernels on the same operands:
    # bit-identical to one stream (test_stage1_chained_weight_gradients_equal_the_two_pass_form); 2.46 -> 2.14 ms per step.
    two_streams = True

    def _prepare(self, B):
        t, b = self.title, self.body
        if B != t.B_alloc:                       # a short last batch: the body rows sit directly behind THIS batch's title rows
            t.extra_rows, t.extra_seqs = B * self.cfg_b.L, B
        t._prepare(B)
        b._prepare(B)

    # ------------------------------------------------------------------ joint passes (round 6)
    joint = True            # False: one launch per pass for everything (the round-5 form; tools/ A/B, tests)
    joint_streams = False   # True: the body's per-pass kernels (embeddings, attention, pooling) on the second stream beside the title's.
                            # Measured (interleaved legs, one box): 1.816 -> 1.718 ms at 30 / 128 and 3.830 -> 3.717 at 24 / 512 WITHOUT it -
                            # ten fork / join pairs per step cost more than running two short kernels side by side wins

    def _joint_ok(self):
        t, b = self.title, self.body
        ok = bool(self.joint and t.drop is None and b.drop is None and self.cfg_t.pooling == "att" and self.dev.type == "cuda"
                  and getattr(t, "fcache", None) is None and t.group_wgrad is False)
        if not ok and getattr(self, "_joint_rows_written", False):
            # a per-pass step after joint ones (dropout switched on, a tools/ A/B): the title engine's own kernels rely on ZERO rows
            # behind its N Lt rows (the weight gradient reads up to the next multiple of 64), where the joint passes have put body
            # rows - lay the title workspace out afresh, once
            t._alloc_workspace(t.B_alloc)
        self._joint_rows_written = ok
        return ok

    def _both(self, body_fn, title_fn):
        """The two passes' own kernels of one stage of the joint passes: side by side on two streams, joined behind."""
        if not self.joint_streams:
            body_fn()
            title_fn()
            return
        main, side = torch.cuda.current_stream(self.dev), self._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            body_fn()
        title_fn()
        main.wait_stream(side)

    def _encode_joint(self, B, N, title_tok, tidx, body_tok, bidx, teacher_side):
        """Engine.encode for both passes at once (model_bert.py:119-137 twice): token rows [0, N Lt) are the titles', [N Lt, N Lt +
        B Lb) the bodies'; news vectors -> S[:N] (titles), S[N:N + B] (bodies)."""
        t, b = self.title, self.body
        cfg = self.cfg_t
        H, D = cfg.H, cfg.D
        Mt = N * cfg.L
        M = Mt + B * self.cfg_b.L
        assert M <= t.Mp and N + B <= t.nv.shape[0]
        g = t.p
        for e in (t, b):
            if e._rel_stale:
                e.refresh_rel()
            e.drop_cur = None
        rows = lambda buf: buf[Mt:M]              # the body pass's rows of a per-token buffer
        self._both(lambda: b._embed_fwd(body_tok, B, bidx, rows(t.x0)),
                   lambda: (teacher_side(), t._embed_fwd(title_tok, N, tidx, t.x0)))
        x = t.x0
        t.x_in = {}
        for l in range(cfg.n_layers):
            names, sh = layer_param_order(l), t.sh[l]
            kept = l >= t.lo
            a = t.act[l - t.lo] if kept else t.scr
            y = a["y"] if kept else t.scr_y[l & 1]
            lse_b = b.act[l - b.lo]["lse"] if kept else b.lse
            t.x_in[l] = x
            t._gemm(x, sh["qkv"], a["qkv"], M, bias=t._view(names[3], 3 * H, (3 * H,)), flags=T.EPI_BIAS)
            self._both(lambda: b._attn_fwd(rows(a["qkv"]), rows(a["ctx"]), lse_b, B, None),
                       lambda: t._attn_fwd(a["qkv"], a["ctx"], a["lse"] if kept else t.lse, N, None))
            t._gemm(a["ctx"], sh["o"], a["h1pre"], M, bias=g(names[7]), res=x, flags=T.EPI_BIAS | T.EPI_RES)
            t._c("tnr_ln_fwd", a["h1pre"], g(names[8]), g(names[9]), cfg.ln_eps, a["h1"], a["st1"], M, H)
            t._gemm(a["h1"], sh["w1"], a["g"], M, bias=g(names[11]), aux=a["u"] if kept else None,
                    flags=T.EPI_BIAS | T.EPI_GELU | (T.EPI_AUXOUT if kept else 0))
            t._gemm(a["g"], sh["w2"], a["ypre"], M, bias=g(names[13]), res=a["h1"], flags=T.EPI_BIAS | T.EPI_RES)
            t._c("tnr_ln_fwd", a["ypre"], g(names[14]), g(names[15]), cfg.ln_eps, y, a["st2"], M, H)
            x = y
        t.y_last = x
        t._gemm(x, t.sh_a1, t.e, M, bias=t.b_a1, flags=T.EPI_BIAS | T.EPI_TANH | T.EPI_OUTF32)
        self._both(lambda: b._attpool_fwd(rows(x), rows(t.e), t.nv[N:N + B], B),
                   lambda: t._attpool_fwd(x, t.e, t.nv, N))
        t._sgemm(t.nv, H, 1, 0, g(PFX + "dense.weight"), H, 1, 0, t.S, D, 0, g(PFX + "dense.bias"), 0, N + B, D, H)

    def _backward_joint(self, dS, B, N, after_bucket, pend, rb_heads, one):
        """Engine.backward_encoder_steps for both passes at once: dS[:N + B] = d loss / d news vectors, titles then bodies."""
        t, b = self.title, self.body
        cfg = self.cfg_t
        D, H, I = cfg.D, cfg.H, cfg.I
        Ns = N + B
        Mt = N * cfg.L
        Mb = B * self.cfg_b.L
        M = Mt + Mb
        g, gr, gi = t.p, t.grads, t.ginv
        t._wg = t._wg_defer = None
        rows = lambda buf: buf[Mt:M]
        seqs = lambda buf: buf[N:Ns]
        rb = rb_heads
        dvec = dS[:Ns]
        t._sgemm_group(list(pend or []) + [
            t._sgemm_problem(dvec, 1, D, 0, t.nv, 1, H, 0, gr[PFX + "dense.weight"], H, 0, None, 0, D, H, Ns, ksplit=t.KS),
            t._sgemm_problem(dvec, D, 1, 0, g(PFX + "dense.weight"), 1, H, 0, t.dnv, H, 0, None, 0, Ns, H, D, alpha=t.gscale)])
        rb.add(dvec, Ns, D, D, gr[PFX + "dense.bias"])
        y = t.y_last
        self._both(lambda: b._attpool_bwd(rows(y), rows(t.e), seqs(t.dnv), rows(t.dy2), rows(t.dpre), seqs(t.dw2p), seqs(t.db2p),
                                          seqs(t.db1p), B),
                   lambda: t._attpool_bwd(y, t.e, t.dnv, t.dy2, t.dpre, t.dw2p, t.db2p, t.db1p, N))
        rb.add(t.dw2p, Ns, cfg.Qn, cfg.Qn, gr[PFX + "attn.att_fc2.weight"], 0, gi)
        rb.add(t.db2p, Ns, 1, 1, gr[PFX + "attn.att_fc2.bias"], 0, gi)
        rb.add(t.db1p, Ns, QPAD, QPAD, t._view(PFX + "attn.att_fc1.bias", QPAD, (QPAD,), grad=True), 0, gi)
        if not one or not cfg.trainable_layers:
            rb.flush()
        t._wgrad(t.dpre, y, t._view(PFX + "attn.att_fc1.weight", QPAD * H, (QPAD, H), grad=True), M)
        if after_bucket:
            after_bucket(0)
        if not cfg.trainable_layers:
            return
        t._gemm(t.dpre, t.sh_a1T, t.dy, M, res=t.dy2, flags=T.EPI_RES)
        dy = t.dy
        bucket = 1
        nblk = T.query("tnr_ln_bwd_blocks", M)
        # q / k / v bias: the short-sequence attention backward leaves one partial row per sequence, the long one none (its dqkv
        # columns are summed into ONE row): both passes' partial rows in one buffer, one reduction job
        nt_rows, nb_rows = (N if cfg.L <= 32 else 1), (B if self.cfg_b.L <= 32 else 1)
        for l in range(cfg.n_layers - 1, t.lo - 1, -1):
            names, sh, a = layer_param_order(l), t.sh[l], t.act[l - t.lo]
            tr = l in cfg.trainable_layers
            x_in = t.x_in[l]
            rba = (rb_heads if one else t.red.setdefault((l, "joint", Ns, "att"), _ReduceBatch(t.dev))) if tr else None
            rb = (rb_heads if one else t.red.setdefault((l, "joint", Ns, "ffn"), _ReduceBatch(t.dev))) if tr else None
            P = t.lpart.get(l)
            t._wg = [] if (tr and self.joint_group_wgrad) else None      # the layer's weight gradients collected for one launch
            t._c("tnr_ln_bwd", dy, a["ypre"], a["st2"], g(names[14]), t.dypre, None, None, None, (P["ln_part"] if tr else None), M, H)
            if tr:
                rb.add(P["ln_part"], nblk, 3 * H, 2 * H, t._view(names[14], 2 * H, (2 * H,), grad=True), 0, gi)
                rb.add(P["ln_part"][2 * H:], nblk, 3 * H, H, gr[names[13]], 0, gi)
                t._wgrad(t.dypre, a["g"], gr[names[12]], M)
            fused_cs = tr and M > 128            # the column-sum epilogue needs more than one 128-row strip (a toy batch has less)
            t._gemm(t.dypre, sh["w2T"], t.du, M, aux=a["u"], flags=T.EPI_MULDGELU | (T.EPI_COLSUM if fused_cs else 0),
                    colsum=P["gcs_part"] if fused_cs else None)
            if tr:
                if fused_cs:
                    rb.add(P["gcs_part"], t._q("tnr_gemm_colsum_rows", M), I, I, gr[names[11]], 0, gi)
                else:
                    t._c("tnr_colsum", t.du, I, T.BF16, M, I, P["cs_tmp"][:I], t.cs_part, 0)
                    rb.add(P["cs_tmp"], 1, I, I, gr[names[11]], 0, gi)
                t._wgrad(t.du, a["h1"], gr[names[10]], M)
                if not one:
                    rb.flush()
                if after_bucket:
                    self._wgrad_flush_joint()          # the FFN block's two gradients: its bucket goes out now
                    after_bucket(bucket)
                    bucket += 1
            t._gemm(t.du, sh["w1T"], t.dh1, M, res=t.dypre, flags=T.EPI_RES)
            t._c("tnr_ln_bwd", t.dh1, a["h1pre"], a["st1"], g(names[8]), t.dh1pre, None, None, None, (P["ln_part1"] if tr else None), M, H)
            if tr:
                rba.add(P["ln_part1"], nblk, 3 * H, 2 * H, t._view(names[8], 2 * H, (2 * H,), grad=True), 0, gi)
                rba.add(P["ln_part1"][2 * H:], nblk, 3 * H, H, gr[names[7]], 0, gi)
                t._wgrad(t.dh1pre, a["ctx"], gr[names[6]], M)
            t._gemm(t.dh1pre, sh["oT"], t.dctx, M)
            qp = P["qkvb_part"] if tr else None

            def body_attn():
                if not b._attn_bwd(rows(a["qkv"]), rows(a["ctx"]), b.act[l - b.lo]["lse"], rows(t.dctx), rows(t.dqkv),
                                   qp[nt_rows:] if tr else None, B, None) and tr:
                    b._c("tnr_colsum", rows(t.dqkv), 3 * H, T.BF16, Mb, 3 * H, qp[nt_rows], b.cs_part, 0)

            def title_attn():
                if not t._attn_bwd(a["qkv"], a["ctx"], a["lse"], t.dctx, t.dqkv, qp, N, None) and tr:
                    t._c("tnr_colsum", t.dqkv, 3 * H, T.BF16, Mt, 3 * H, qp[0], t.cs_part, 0)
            self._both(body_attn, title_attn)
            if tr:
                rba.add(qp, nt_rows + nb_rows, 3 * H, 3 * H, t._view(names[3], 3 * H, (3 * H,), grad=True), 0, gi)
                t._wgrad(t.dqkv, x_in, t._view(names[0], 3 * H * H, (3 * H, H), grad=True), M)
                self._wgrad_flush_joint()              # before the next layer overwrites their operands
                if not one:
                    rba.flush()
            if l > t.lo:
                nxt = t.dy2 if dy is t.dy else t.dy
                t._gemm(t.dqkv, sh["qkvT"], nxt, M, res=t.dh1pre, flags=T.EPI_RES)
                dy = nxt
            if tr and after_bucket:
                after_bucket(bucket)
                bucket += 1
        t._wg = None
        if one:
            rb_heads.flush()

    joint_group_wgrad = True    # the joint passes' weight gradients of a layer in ONE persistent launch + one slab sum

    def _wgrad_flush_joint(self):
        """The weight gradients collected since the last flush (a layer's four; two and two under a bucket hook) as one
        tnr_gemm_tn_wgrad_group launch.  The joint passes have few rows (8 896 at 30 / 128): a launch per gradient with a full
        round of (split, tile) units each (Engine._wgrad_splits) means 20 m steps per unit, 7 fp32 slabs per gradient and a slab
        sum per gradient; here ALL the collected gradients share one round - every gradient gets the same number of splits, so
        every unit runs the same number of m steps - and one slab sum."""
        t = self.title
        pend, t._wg = t._wg, ([] if t._wg is not None else None)
        if not pend:
            return
        tiles = sum((N // 256) * (K // 256) for _, _, _, _, N, K, _ in pend)
        assert all(N % 256 == 0 and K % 256 == 0 for _, _, _, _, N, K, _ in pend)
        splits = max(1, min(64, (256 * Engine.WGRAD_UNITS) // tiles))
        probs, off = [], 0
        for dy, x, dw, M, N, K, acc in pend:
            elems = T.query("tnr_gemm_tn_ws_elems", N, K, splits)
            probs.append(dict(dY=dy, lddy=dy.stride(0), X=x, ldx=x.stride(0), dW=dw, lddw=dw.stride(0), M=M, N=N, K=K,
                              ws=t.ws[off:off + elems], splits=splits, accumulate=acc, out_scale=t.ginv))
            off += (elems + 63) // 64 * 64
        assert off <= t.ws.numel() and len(probs) <= 4
        T.wgrad_group(probs, f16=t.f16)

    def _encode_both(self, body_pass, title_pass):
        if not self.two_streams or self.dev.type != "cuda":
            body_pass()
            title_pass()
            return
        main, side = torch.cuda.current_stream(self.dev), self._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            body_pass()
        title_pass()
        main.wait_stream(side)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.dev)
        return self._side

    def _teacher_side(self, B, N, Rt):
        """Teacher scores <title_emb[b, c], body_emb[b]> and the teachers' projections from the teacher rows X: independent fp32
        GEMMs of a few microseconds each - one grouped launch (each computed exactly as by its own tnr_sgemm call)."""
        t = self.title
        C, D, T_ = self.cfg_t.C, self.cfg_t.D, self.cfg_t.T
        X = t.X
        group = [t._sgemm_problem(X[i], D, 1, C * D, X[i, N:], D, 1, D, t.t_score[i], 1, C, None, 0, C, 1, D, batch=B) for i in range(T_)]
        Wt = t._view("transform_matrix.0.weight", T_ * D * D, (T_, D, D))
        bt = t._view("transform_matrix.0.bias", T_ * D, (T_, D))
        group.append(t._sgemm_problem(X, D, 1, X.stride(0), Wt, D, 1, D * D, t.Pm, D, t.Pm.stride(0), bt, D, Rt, D, D, batch=T_))
        for i in range(0, len(group), 8):
            t._sgemm_group(group[i:i + 8])

    def _heads(self, B, N, Rt):
        """Scores, teacher weights and the three losses from the student rows S and the teacher rows X."""
        t = self.title
        cfg = self.cfg_t
        C, D, T_ = cfg.C, cfg.D, cfg.T
        S = t.S[:Rt]
        # score[b, c] = <title_vec[b, c], body_vec[b]>
        t._sgemm(S, D, 1, C * D, S[N:], D, 1, D, t.score, 1, C, None, 0, C, 1, D, batch=B)
        T.call("tnr_kd_score_loss", t.score, t.t_score if T_ else None, t.label, 1.0, 1.0, t.tw if T_ else None, t.dscore,
               t.losses, B, C, T_)
        if T_:
            T.call("tnr_kd_embed_loss", S, t.Pm, t.tw, t.losses[2:], t.dS, t.dP, t.kd_part, B, 0, C, D, T_)
        else:
            t.dS[:Rt].zero_()
            t.losses[2:3].zero_()
        return t.losses, t.score[:B]

    @torch.no_grad()
    def encode_table(self, table, which):
        """Embeddings of every row of a resident token table (Domian-specific_Post-train.ipynb cells 20-22: the teacher
        title / body embeddings stage 1 distils from).  which: "title" | "body".  -> (n, D) fp32 on device."""
        eng = self.title if which == "title" else self.body
        assert table.shape[1] == 2 * eng.cfg.L
        return eng.encode_news(table)

    def total_loss(self):
        """target + distill + emb (cell 14) as a device scalar."""
        l = self.title.losses
        return l[0] + l[1] + l[2]

    def backward(self, after_bucket=None):
        """Gradients of total_loss -> the shared flat_g.  Buckets complete (and after_bucket fires) during the body
        pass, the second and accumulating one."""
        t, b = self.title, self.body
        t._red_check()
        b._red_check()
        B, N, Rt = self.cur
        C, D = self.cfg_t.C, self.cfg_t.D
        S, dS = t.S[:Rt], t.dS
        # the title pass's heads batch (backward_encoder's key): merged into one reduction at its end unless it flushes by bucket
        pend = []                 # the transform matrices' gradient GEMM rides in the title pass's first grouped launch
        if self.ran_joint:
            one = after_bucket is None and t.merge_reductions
            rbh = t.red.setdefault(("heads", "joint", N + B, one), _ReduceBatch(t.dev))
            if self.cfg_t.T:
                t._transform_grads(Rt, rbh, pend)
            T.call("tnr_score_bwd", S, t.cidx, S[N:], t.dscore, dS, dS[N:], B, C, D)
            self._backward_joint(dS, B, N, after_bucket, pend, rbh, one)
            return
        one_t = t.merge_reductions and not (self.chain_wgrad and after_bucket is not None)
        if self.cfg_t.T:
            t._transform_grads(Rt, t.red.setdefault(("heads", 0, N, one_t), _ReduceBatch(t.dev)), pend)
        T.call("tnr_score_bwd", S, t.cidx, S[N:], t.dscore, dS, dS[N:], B, C, D)
        if not self.chain_wgrad:
            t.backward_encoder(dS[:N], N, acc=0, pend=pend)
            b.backward_encoder(dS[N:Rt], B, acc=1, after_bucket=after_bucket)
            return
        # The two passes' backwards in step, layer by layer: every shared weight gets ONE chained weight-gradient problem (title
        # rows, then body rows; one fixed-order slab sum) instead of a launch + slab sum that writes and a launch + slab sum that
        # adds.  The title's segment runs first, so wherever both flush partial sums the writing flush precedes the adding one;
        # under a bucket hook the title pass flushes bucket by bucket as well (a hook of its own that does nothing).
        hooked = after_bucket is not None
        gt = t.backward_encoder_steps(dS[:N], N, acc=0, after_bucket=(lambda i: None) if hooked else None, defer=True, split_ffn=hooked,
                                      pend=pend)
        gb = b.backward_encoder_steps(dS[N:Rt], B, acc=1, after_bucket=after_bucket, defer=True, split_ffn=hooked)
        # side by side only when BOTH passes hold their partial sums back for one merged reduction at the very end: with
        # Engine.merge_reductions off (tools/step_ab.py) each pass flushes inside its own segments, and the title's writing flush
        # (main stream) would be unordered against the body's adding flush (side stream) on the shared bias / LayerNorm gradients
        if not (self.two_streams and not hooked and self.dev.type == "cuda" and t.merge_reductions and b.merge_reductions):
            while True:
                at, ab = next(gt, None), next(gb, None)
                assert at == ab, (at, ab)
                if at is None:
                    break
                self._wgrad_flush_chained()
            return
        # ... and side by side: the body pass's segments on the second stream.  What the two passes share are the gradients: the
        # weights' are written by the chained launches (main stream, both passes joined in front of it, released behind it); all
        # others go through the two passes' batched reductions, which without a bucket hook run once, at the very end - the
        # title's (writing) and then the body's (adding), both on the main stream behind the join; the one gradient written
        # directly - dense.weight, first thing in the heads segment - orders the body's heads segment behind the title's.
        main, side = torch.cuda.current_stream(self.dev), self._side_stream()
        side.wait_stream(main)
        first = True
        while True:
            at = next(gt, None)
            if at is None:                      # the title's merged reduction has been launched: the body's follows on the same stream
                main.wait_stream(side)
                assert next(gb, None) is None
                break
            if first:
                side.wait_stream(main)
                first = False
            with torch.cuda.stream(side):
                ab = next(gb, None)
            assert at == ab, (at, ab)
            main.wait_stream(side)
            self._wgrad_flush_chained()
            side.wait_stream(main)

    chain_wgrad = True      # False: the title pass writes, the body pass accumulates (two launches + two slab sums per weight)

    def _wgrad_flush_chained(self):
        t, b = self.title, self.body
        pt, pb = t._wg_defer, b._wg_defer
        assert len(pt) == len(pb) and t.ginv == b.ginv
        if self.ws is None or self.ws.numel() < 2 * t.ws.numel():
            self.ws = torch.zeros(2 * t.ws.numel() + 256, device=self.dev, dtype=torch.float32)
        probs, off = [], 0
        for (dy1, x1, dw, M1, N_, K_, acc1), (dy2, x2, dw2, M2, N2, K2, acc2) in zip(pt, pb):
            assert dw.data_ptr() == dw2.data_ptr() and (N_, K_) == (N2, K2) and acc1 == 0 and acc2 == 1
            total = t._wgrad_splits(N_, K_)[0]                   # the units of one round, shared out by rows
            s1 = min(max(1, round(total * M1 / (M1 + M2))), max(1, total - 1))
            s2 = max(1, total - s1)
            elems = (s1 + s2) * N_ * K_
            common = dict(dW=dw, lddw=dw.stride(0), N=N_, K=K_, out_scale=t.ginv)
            probs.append(dict(common, dY=dy1, lddy=dy1.stride(0), X=x1, ldx=x1.stride(0), M=M1, ws=self.ws[off:off + elems], splits=s1,
                              accumulate=0))
            probs.append(dict(common, dY=dy2, lddy=dy2.stride(0), X=x2, ldx=x2.stride(0), M=M2, ws=None, splits=s2, accumulate=2))
            off += (elems + 63) // 64 * 64
            assert off <= self.ws.numel()
        for i in range(0, len(probs), 4):
            T.wgrad_group(probs[i:i + 4], f16=t.f16)
        del pt[:], pb[:]

    def bucket_ranges(self):
        return self.title.bucket_ranges()

    def step(self, lr, grad_scale=1.0, lr_bert=None, amsgrad=False, **kw):
        """Post-train_KD.ipynb cell 18: optim.Adam([{bert_model, 1e-6}, {rest, 1e-5}]) (plain Adam by default here)."""
        self.title.step(lr, grad_scale, lr_bert=lr_bert, amsgrad=amsgrad, **kw)
