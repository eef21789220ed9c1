"""Training-step engine: chains the libtnr_hip.so kernels into forward / backward / AMSGrad.

Host-side mirror of the reference's hot loop body (run.py:178-195): Model.forward
(model_bert.py:262-305) -> total_loss.backward() -> optimizer.step(), for one worker's batch.
torch is used for device memory, streams and (in dist.py) the RCCL all-reduce only; every
arithmetic op on the path is a HIP kernel behind the C ABI, and there is no fallback.

Memory plan (sized for 288 GB HBM3E): parameters live in two flat fp32 buffers (trainable /
frozen) whose slices ARE the nn.Parameters of model_bert.py, gradients and AMSGrad state are flat
buffers of the trainable size, bf16 (and transposed bf16) weight copies are refreshed by one
kernel after each update, activations of the trainable layers stay resident for backward.
"""
import logging
import math
import struct

import torch

import tnr_hip as T

PFX = "student.news_encoder."
BERT = PFX + "bert_model.bert."
QPAD = 256          # news pooling query dim (200) padded to the GEMM tile
AP_LONG = 64        # sequences longer than this take the chunked pooling kernels (tnr_attpool_*_long); a rule of L alone


def _rup(x, m):
    return (x + m - 1) // m * m


class EngineConfig:
    def __init__(self, n_layers=4, trainable_layers=(2, 3), hidden=768, heads=12, inter=3072, news_dim=256,
                 news_query=200, user_query=200, num_teachers=4, user_log_length=50, npratio=4, num_words=30,
                 user_log_mask=False, temperature=1.0, coef=0.2, vocab=30522, max_pos=512, type_vocab=2,
                 ln_eps=1e-12, stage1=False, pooling="att", nrms_heads=0, pos_pad_id=None):
        """stage1=True: the DistillModel of Post-train_KD.ipynb (no user encoders: parameters are
        student.news_encoder.* and transform_matrix.* only; user_log_length is 0, npratio+1 titles per body)."""
        self.stage1 = stage1
        # PLM-NR --model_type (PLM-NR/utils.py:17-21): 'bert' is this encoder with a zero rel-pos table (rel_pos_bias.weight = 0);
        # 'roberta' additionally takes its position rows from the token ids -- cumulative count of non-pad tokens + padding_idx
        # (transformers create_position_ids_from_input_ids) -- with type_vocab 1, max_pos 514 and ln_eps 1e-5 from its config
        self.pos_pad_id = pos_pad_id
        # args.pooling (model_bert.py:130-135): 'att' | 'cls' | anything else = mean ; args.model == 'NRMS' puts a
        # nrms_heads x 16 self-attention in front of every user encoder's pooling (model_bert.py:145-148)
        self.pooling = pooling if pooling in ("att", "cls") else "mean"
        self.nrms_heads = int(nrms_heads)
        assert not self.nrms_heads or self.nrms_heads * 16 == news_dim, "NRMS: num_attention_heads * 16 must equal news_dim (scorer bmm)"
        self.n_layers, self.trainable_layers = n_layers, tuple(sorted(trainable_layers))
        self.H, self.A, self.I, self.D = hidden, heads, inter, news_dim
        self.Qn, self.Qu, self.T = news_query, user_query, num_teachers
        self.U, self.C, self.L = user_log_length, npratio + 1, num_words
        self.user_log_mask, self.temperature, self.coef = bool(user_log_mask), float(temperature), float(coef)
        self.vocab, self.max_pos, self.type_vocab, self.ln_eps = vocab, max_pos, type_vocab, ln_eps
        assert hidden == heads * 64, "attention kernel is built for head size 64"
        assert hidden % 256 == 0 and inter % 128 == 0 and news_dim % 4 == 0 and news_query <= QPAD
        assert 1 <= num_words <= 512, "attention kernels cover sequences of up to 512 tokens"
        assert all(0 <= l < n_layers for l in self.trainable_layers)


NRMS_W = ["multi_head_self_attn.W_%s.weight" % n for n in "QKV"]      # stacked [W_Q; W_K; W_V] (3D, D) per encoder
NRMS_B = ["multi_head_self_attn.W_%s.bias" % n for n in "QKV"]


def layer_param_order(l):
    p = BERT + "encoder.layer.%d." % l
    return [p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight",
            p + "attention.self.query.bias", p + "attention.self.key.bias", p + "attention.self.value.bias",
            p + "attention.output.dense.weight", p + "attention.output.dense.bias",
            p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
            p + "intermediate.dense.weight", p + "intermediate.dense.bias",
            p + "output.dense.weight", p + "output.dense.bias",
            p + "output.LayerNorm.weight", p + "output.LayerNorm.bias"]


def param_shapes(cfg):
    """state_dict schema of model_bert.Model, in the engine's storage order."""
    H, I, D, T_ = cfg.H, cfg.I, cfg.D, cfg.T
    s = {}
    if cfg.nrms_heads and not cfg.stage1:
        for i in range(T_):
            for n in NRMS_W:
                s["teachers.%d.%s" % (i, n)] = (D, D)
        for i in range(T_):
            for n in NRMS_B:
                s["teachers.%d.%s" % (i, n)] = (D,)
    for i in range(0 if cfg.stage1 else T_):
        s["teachers.%d.attn.att_fc1.weight" % i] = (cfg.Qu, D)
    for i in range(0 if cfg.stage1 else T_):
        s["teachers.%d.attn.att_fc1.bias" % i] = (cfg.Qu,)
    for i in range(0 if cfg.stage1 else T_):
        s["teachers.%d.attn.att_fc2.weight" % i] = (1, cfg.Qu)
    for i in range(0 if cfg.stage1 else T_):
        s["teachers.%d.pad_doc" % i] = (1, D)
    for i in range(0 if cfg.stage1 else T_):
        s["teachers.%d.attn.att_fc2.bias" % i] = (1,)
    s[BERT + "embeddings.word_embeddings.weight"] = (cfg.vocab, H)
    s[BERT + "embeddings.position_embeddings.weight"] = (cfg.max_pos, H)
    s[BERT + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab, H)
    s[BERT + "embeddings.LayerNorm.weight"] = (H,)
    s[BERT + "embeddings.LayerNorm.bias"] = (H,)
    for l in range(cfg.n_layers):
        for k in layer_param_order(l):
            if k.endswith("intermediate.dense.weight"):
                s[k] = (I, H)
            elif k.endswith("intermediate.dense.bias"):
                s[k] = (I,)
            elif k.endswith(".output.dense.weight") and "attention" not in k:
                s[k] = (H, I)
            elif k.endswith("weight") and "LayerNorm" not in k:
                s[k] = (H, H)
            else:
                s[k] = (H,)
    s[BERT + "pooler.dense.weight"] = (H, H)
    s[BERT + "pooler.dense.bias"] = (H,)
    s[BERT + "rel_pos_bias.weight"] = (cfg.A, 32)
    s[PFX + "bert_model.classifier.weight"] = (2, H)
    s[PFX + "bert_model.classifier.bias"] = (2,)
    # heads: order = order in which their gradients complete / are laid out by the kernels
    if cfg.pooling == "att":
        s[PFX + "attn.att_fc1.weight"] = (cfg.Qn, H)
        s[PFX + "attn.att_fc1.bias"] = (cfg.Qn,)
        s[PFX + "attn.att_fc2.weight"] = (1, cfg.Qn)
        s[PFX + "attn.att_fc2.bias"] = (1,)
    s[PFX + "dense.weight"] = (D, H)
    s[PFX + "dense.bias"] = (D,)
    if cfg.nrms_heads and not cfg.stage1:
        for n in NRMS_W:
            s["student.user_encoder." + n] = (D, D)
        for n in NRMS_B:
            s["student.user_encoder." + n] = (D,)
    if not cfg.stage1:
        s["student.user_encoder.attn.att_fc1.weight"] = (cfg.Qu, D)
        s["student.user_encoder.attn.att_fc1.bias"] = (cfg.Qu,)
        s["student.user_encoder.attn.att_fc2.weight"] = (1, cfg.Qu)
        s["student.user_encoder.pad_doc"] = (1, D)
        s["student.user_encoder.attn.att_fc2.bias"] = (1,)
    for i in range(T_):
        s["transform_matrix.%d.weight" % i] = (D, D)
    for i in range(T_):
        s["transform_matrix.%d.bias" % i] = (D,)
    return s


def is_trainable(cfg, name):
    """run.py:101-112: teachers frozen; bert_model frozen except encoder.layer[i], i in trainable_layers."""
    if name.startswith("teachers."):
        return False
    if name.startswith(PFX + "bert_model."):
        for l in cfg.trainable_layers:
            if name.startswith(BERT + "encoder.layer.%d." % l):
                return True
        return False
    return True


class _ReduceBatch:
    """Fixed-order partial-sum reductions of one gradient bucket, done by ONE tnr_reduce_multi launch.
    The job list is static (same buffers every step): recorded during the first backward, then replayed."""

    def __init__(self, dev):
        self.dev, self.jobs, self.table = dev, [], None

    def add(self, part, rows, stride, n, out, accumulate=0, scale=1.0):
        """out (+)= scale * column sums of part ; the scale rides in the upper half of the descriptor's last word."""
        if self.table is None:
            sb = 0 if scale == 1.0 else struct.unpack("<I", struct.pack("<f", scale))[0]
            self.jobs.append((part.data_ptr(), rows, stride, n, out.data_ptr(), int(accumulate) | (sb << 32)))

    def flush(self):
        if not self.jobs:
            return
        if self.table is None:
            l1, l2 = [], []
            for part, rows, stride, n, out, acc in self.jobs:
                chunks = max(1, min(rows // 32, 16))                 # row chunks summed in place first (tall partials)
                per = (rows + chunks - 1) // chunks
                chunks = (rows + per - 1) // per
                for c0 in range(0, n, 64):
                    nc = min(64, n - c0)
                    if chunks > 1:
                        for ch in range(chunks):
                            r0 = ch * per
                            src = part + 4 * (r0 * stride + c0)
                            l1.append((src, min(per, rows - r0), stride, nc, src, 0))
                        l2.append((part + 4 * c0, chunks, per * stride, nc, out + 4 * c0, acc))
                    else:
                        l2.append((part + 4 * c0, rows, stride, nc, out + 4 * c0, acc))
            mk = lambda rows_: (torch.tensor(rows_, dtype=torch.int64, device=self.dev), len(rows_)) if rows_ else None
            self.table = (mk(l1), mk(l2))
        for t in self.table:
            if t is not None:
                T.call("tnr_reduce_multi", t[0], t[1])


LOSS_SCALE = 1024.0     # static scale of the 16-bit backward in fp16 mode (gradients of ~1e-6 would underflow); it enters at
                        # the pooling backward and leaves in the kernels that write parameter gradients (weight-gradient slab
                        # reduce, bias / LayerNorm partial reductions), so flat_g always holds the true gradients


class LossScaler:
    """Dynamic loss scale of the fp16 build (host side; the device side is tnr_grad_nonfinite + tnr_amsgrad_step_guarded).
    The reference trains in fp32 (run.py:134,194-195) and has no such thing; with 16-bit activation gradients ONE overflow would
    put inf / nan into m, v and - for good - vmax.  Every optimiser step gets a stamp; a kernel looks for non-finite values in
    the (reduced) flat gradient and records the stamp in `guard`, and the update kernels of that stamp do nothing: the step is
    skipped on the device without the host knowing.  The host reads guard back asynchronously and looks at the answer TWO steps
    later (the GPU is never drained for it): an overflow halves the multiplier `mult` (the engine's scale is base x mult) and takes
    the skipped step out of Adam's step count; `growth_interval` clean steps in a row double it again, up to `max_mult`.
    bf16 has fp32's exponent range: scaler disabled, nothing launched."""

    def __init__(self, dev, enabled, growth_interval=2000, max_mult=64.0, min_mult=2.0 ** -20):
        self.enabled = bool(enabled) and torch.device(dev).type == "cuda"
        self.mult, self.stamp, self.clean, self.skipped, self.run_of_skips = 1.0, 0, 0, 0, 0
        self.growth_interval, self.max_mult, self.min_mult = growth_interval, max_mult, min_mult
        self.guard = torch.zeros(4, dtype=torch.int32, device=dev) if self.enabled else None
        self.pending = []                    # (stamp, pinned host word, event, multiplier that step's backward ran with) in issue order
        self._ring = torch.zeros(8, dtype=torch.int32).pin_memory() if self.enabled else None    # at most lag + 1 answers are outstanding

    def poll(self, eng, lag=2):
        """Answers of the steps at least `lag` stamps back.  Deterministic (which answers are used depends on the step number
        only, never on timing: data-parallel ranks must change their scale and step count at the same step); the event of a step
        two back has normally long completed - a host running further ahead than that waits here, with a full step still queued.
        An answer arrives two steps late, so the two backwards after an overflow have already run at the old scale and
        overflow with it: an overflow takes the multiplier to HALF OF WHAT THAT STEP'S BACKWARD RAN WITH, if it is not below
        that already - one cause, one halving; the later two steps are still skipped on the device and still leave Adam's
        count."""
        while self.pending and self.pending[0][0] <= self.stamp - lag + 1:
            stamp, host, ev, used = self.pending.pop(0)
            ev.synchronize()
            if int(host[0]) == stamp:        # that step found inf / nan: it was skipped on the device
                self.skipped += 1
                self.clean = 0
                self.run_of_skips += 1
                self.mult = max(min(self.mult, used * 0.5), self.min_mult)
                eng.step_count = max(eng.step_count - 1, 0)
                if self.run_of_skips == 16:  # a smaller scale cures a backward overflow within a few steps, never a forward one
                    logging.warning("16 optimiser steps in a row skipped for inf / nan gradients although the loss scale went down "
                                    "to %g: the fp16 FORWARD overflows (or the data holds inf / nan) - use --dtype bf16", eng.gscale)
            else:
                self.run_of_skips = 0
                self.clean += 1
                if self.clean >= self.growth_interval and self.mult < self.max_mult:
                    self.mult, self.clean = self.mult * 2.0, 0

    def record(self, used):
        """`used`: the multiplier this step's backward ran with (Engine.step reads it before poll may change it)."""
        host = self._ring[self.stamp % 8: self.stamp % 8 + 1]
        host.copy_(self.guard[:1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((self.stamp, host, ev, used))

    def drain(self, eng):
        """Wait for every outstanding answer (end of an epoch / before a checkpoint / tests)."""
        self.poll(eng, lag=0)

    def reset(self, eng):
        """New weights (load_state_dict) or a new benchmark leg: outstanding answers are consumed, then the scale starts over;
        skipped-step totals stay (they are a log figure)."""
        self.drain(eng)
        self.mult, self.clean, self.run_of_skips = 1.0, 0, 0


class Engine:
    def __init__(self, cfg, device="cuda:0", max_batch=32, dtype="fp16", share=None, extra_rows=0, extra_seqs=0):
        """extra_rows / extra_seqs: room for that many more token rows / sequences behind this engine's own in every per-token /
        per-sequence workspace (stage 1's joint passes: the body pass's rows directly behind the title pass's, stage1.py).
        dtype: 16-bit activation / weight-copy type, "fp16" (default: meets the 1e-3 logit / loss bound) or "bf16".
        fp16 has the same MFMA rate and 3 more mantissa bits; its backward runs on activation gradients scaled by LOSS_SCALE from the
        pooling backward down (everything 16-bit); the kernels that write parameter gradients multiply by 1 / LOSS_SCALE (exact)."""
        T.lib()                      # fail loudly if the HIP library is missing
        assert dtype in ("bf16", "fp16")
        self.f16 = dtype == "fp16"
        self.dtype = dtype
        self.tdt = torch.float16 if self.f16 else torch.bfloat16
        self._gbase = LOSS_SCALE if self.f16 else 1.0
        self.cfg, self.dev = cfg, torch.device(device)
        # dynamic loss scale (fp16 only); engines that share parameters share it (stage 1: both passes of a step use one scale)
        self.scaler = share.scaler if share is not None else LossScaler(self.dev, self.f16)
        self.step_count = 0
        self._n_alloc = 0
        if share is None:
            self._build_params()
            self._build_shadows()
        else:
            # a second pass over the SAME model at another sequence length (stage 1: titles and bodies): parameters,
            # gradients, optimiser state and 16-bit weight copies are the other engine's, workspaces are its own
            for k in ("shapes", "slot", "n_train", "n_frozen", "flat", "flat_g", "adam_m", "adam_v", "adam_vmax", "params",
                      "grads", "lo", "sh", "sh_a1", "sh_a1T", "b_a1", "desc_all", "desc_train"):
                setattr(self, k, getattr(share, k))
        self.max_batch = max_batch
        self.extra_rows, self.extra_seqs = int(extra_rows), int(extra_seqs)
        self.drop = None             # set_dropout(): train-mode dropout of the stage-0 / stage-1 notebooks
        self.drop_calls = 0
        self._alloc_workspace(max_batch)
        self.comm = None             # set by dist.attach()

    @property
    def gscale(self):
        """Loss scale of the 16-bit backward: LOSS_SCALE x (batch factor) x the dynamic multiplier (powers of two: exact)."""
        return self._gbase * self.scaler.mult

    @property
    def ginv(self):
        return 1.0 / self.gscale

    # ------------------------------------------------------------------ parameters
    def _groups(self):
        """Storage groups: members are contiguous (no padding inside), every group starts 64-float aligned.
        Stacks that kernels read as one array: per-teacher params, q/k/v weights and biases, the student
        user-encoder block (w1, then tnr_user_bwd_pre/post's partial layout [b1|w2|pad|b2]), transform matrices."""
        cfg = self.cfg
        T_ = cfg.T
        tstack = lambda suffix: ["teachers.%d.%s" % (i, suffix) for i in range(T_)]
        groups = []
        if T_ and cfg.nrms_heads and not cfg.stage1:
            groups.append((["teachers.%d.%s" % (i, n) for i in range(T_) for n in NRMS_W], None))
            groups.append((["teachers.%d.%s" % (i, n) for i in range(T_) for n in NRMS_B], None))
        if T_ and not cfg.stage1:
            for suffix in ("attn.att_fc1.weight", "attn.att_fc1.bias", "attn.att_fc2.weight", "pad_doc", "attn.att_fc2.bias"):
                groups.append((tstack(suffix), None))
        for k in ("word_embeddings.weight", "position_embeddings.weight", "token_type_embeddings.weight",
                  "LayerNorm.weight", "LayerNorm.bias"):
            groups.append(([BERT + "embeddings." + k], None))
        for l in range(cfg.n_layers):
            n = layer_param_order(l)
            groups.append((n[0:3], None))
            groups.append((n[3:6], None))
            for k in n[6:]:
                groups.append(([k], None))
        for k in (BERT + "pooler.dense.weight", BERT + "pooler.dense.bias", BERT + "rel_pos_bias.weight",
                  PFX + "bert_model.classifier.weight", PFX + "bert_model.classifier.bias"):
            groups.append(([k], None))
        if cfg.pooling == "att":
            groups.append(([PFX + "attn.att_fc1.weight"], QPAD * cfg.H))     # rows Qn..QPAD stay zero
            groups.append(([PFX + "attn.att_fc1.bias"], QPAD))
            for k in ("attn.att_fc2.weight", "attn.att_fc2.bias"):
                groups.append(([PFX + k], None))
        for k in ("dense.weight", "dense.bias"):
            groups.append(([PFX + k], None))
        ue = "student.user_encoder."
        if cfg.nrms_heads and not cfg.stage1:
            groups.append(([ue + n for n in NRMS_W], None))
            groups.append(([ue + n for n in NRMS_B], None))
        if not cfg.stage1:
            groups.append(([ue + "attn.att_fc1.weight", ue + "attn.att_fc1.bias", ue + "attn.att_fc2.weight", ue + "pad_doc",
                            ue + "attn.att_fc2.bias"], None))
        if T_:
            groups.append((["transform_matrix.%d.weight" % i for i in range(T_)], None))
            groups.append((["transform_matrix.%d.bias" % i for i in range(T_)], None))
        return groups

    def _build_params(self):
        cfg = self.cfg
        shapes = param_shapes(cfg)
        self.shapes = shapes
        off = {True: 0, False: 0}
        self.slot = {}
        seen = set()
        for names, reserve in self._groups():
            tr = is_trainable(cfg, names[0])
            o = off[tr]
            for name in names:
                shp = shapes[name]
                n = int(math.prod(shp))
                self.slot[name] = (tr, o, n, shp)
                o += n
                seen.add(name)
            off[tr] = _rup(max(o, off[tr] + (reserve or 0)), 64)
        assert seen == set(shapes), set(shapes) ^ seen
        self.n_train, self.n_frozen = max(off[True], 64), max(off[False], 64)
        dev = self.dev
        self.flat = {True: torch.zeros(self.n_train, device=dev), False: torch.zeros(self.n_frozen, device=dev)}
        self.flat_g = torch.zeros(self.n_train, device=dev)
        self.adam_m = torch.zeros(self.n_train, device=dev)
        self.adam_v = torch.zeros(self.n_train, device=dev)
        self.adam_vmax = torch.zeros(self.n_train, device=dev)
        self.params, self.grads = {}, {}
        for name in shapes:
            tr, o, n, shp = self.slot[name]
            self.params[name] = self.flat[tr][o:o + n].view(shp)
            if tr:
                self.grads[name] = self.flat_g[o:o + n].view(shp)

    def _c(self, name, *args):
        """Kernel call; 16-bit entry points take the _f16 variant in fp16 mode."""
        T.call(name + "_f16" if (self.f16 and name in T.TYPED) else name, *args)

    def _q(self, name, *args):
        return T.query(name + "_f16" if (self.f16 and name in T.TYPED) else name, *args)

    def p(self, name):
        return self.params[name]

    def off(self, name):
        return self.slot[name][1]

    def _view(self, first, numel, shape, grad=False):
        tr, o, _, _ = self.slot[first]
        buf = self.flat_g if grad else self.flat[tr]
        return buf[o:o + numel].view(shape)

    def load_state_dict(self, sd):
        """sd: {key: array-like fp32}; every key of the reference schema must be present."""
        missing = [k for k in self.shapes if k not in sd]
        if missing:
            raise KeyError("missing keys: %s" % missing[:5])
        for k, shp in self.shapes.items():
            v = sd[k]
            v = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
            if tuple(v.shape) != tuple(shp):
                raise ValueError("%s: shape %s != %s" % (k, tuple(v.shape), shp))
            self.params[k].copy_(v.to(self.dev, torch.float32))
        self.refresh_shadows(all_layers=True)
        if self.scaler.enabled:              # answers still outstanding belong to the old weights; the scale starts over
            self.scaler.reset(self)

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.params.items()}

    # ------------------------------------------------------------------ dropout (train mode of the notebooks)
    def set_dropout(self, p_hidden, p_attn, seed, pass_id=0):
        """hidden_dropout_prob / attention_probs_dropout_prob of the UniLM config (tnlrv3/config/*.json:2,4), live in the
        notebooks because they call .train() (Post-train_KD.ipynb cell 19:6): embeddings (tnlrv3/modeling.py:177), attention
        probabilities (:224), BertSelfOutput / BertOutput (:287, :306).  run.py never calls .train(), so its path has none.
        Counter-based masks (csrc/dropout.h): every encode() is one numbered forward call = 2 * count + pass_id (two engines that
        share parameters -- titles and bodies -- take different pass_ids); backward regenerates the masks from that number.
        p = 0 for both turns dropout off (bit-identical to never having called this)."""
        if (p_hidden or 0.0) <= 0.0 and (p_attn or 0.0) <= 0.0:
            self.drop = None
            return
        self.drop = dict(p_hidden=float(p_hidden or 0.0), p_attn=float(p_attn or 0.0), seed=int(seed), pass_id=int(pass_id))
        if not hasattr(self, "dyprem"):
            z = lambda *s_: torch.zeros(s_, device=self.dev, dtype=self.tdt)
            self.dyprem, self.dh1prem = z(self.Mp, self.cfg.H), z(self.Mp, self.cfg.H)

    def _dsite(self, kind, layer):
        """tnr_dropout_t of one site of the CURRENT forward call (None when dropout is off for that kind)."""
        d = self.drop_cur
        if d is None:
            return None
        return T.Dropout.site_of(d["p_attn"] if kind == T.DROP_PROB else d["p_hidden"], d["seed"], kind, layer, d["call"])

    # ------------------------------------------------------------------ bf16 weight copies
    def _build_shadows(self):
        cfg, dev, bf = self.cfg, self.dev, self.tdt
        H, I = cfg.H, cfg.I
        lo = min(cfg.trainable_layers) if cfg.trainable_layers else cfg.n_layers
        self.lo = lo
        self.sh = []
        for l in range(cfg.n_layers):
            d = dict(qkv=torch.zeros((3 * H, H), device=dev, dtype=bf), o=torch.zeros((H, H), device=dev, dtype=bf),
                     w1=torch.zeros((I, H), device=dev, dtype=bf), w2=torch.zeros((H, I), device=dev, dtype=bf))
            if l >= lo:      # dgrad needs the transposed copies
                d.update(qkvT=torch.zeros((H, 3 * H), device=dev, dtype=bf), oT=torch.zeros((H, H), device=dev, dtype=bf),
                         w1T=torch.zeros((H, I), device=dev, dtype=bf), w2T=torch.zeros((I, H), device=dev, dtype=bf))
            self.sh.append(d)
        self.sh_a1 = torch.zeros((QPAD, H), device=dev, dtype=bf)
        self.sh_a1T = torch.zeros((H, QPAD), device=dev, dtype=bf)
        self.b_a1 = self._view(PFX + "attn.att_fc1.bias", QPAD, (QPAD,)) if cfg.pooling == "att" else None

        def table(layers, with_heads):
            rows = []
            for l in layers:
                d, names = self.sh[l], layer_param_order(l)
                for i in range(3):
                    rows.append((self.p(names[i]), H, H, d["qkv"][i * H:], H, d["qkvT"][:, i * H:] if "qkvT" in d else None, 3 * H))
                rows.append((self.p(names[6]), H, H, d["o"], H, d.get("oT"), H))
                rows.append((self.p(names[10]), I, H, d["w1"], H, d.get("w1T"), I))
                rows.append((self.p(names[12]), H, I, d["w2"], I, d.get("w2T"), H))
            if with_heads and cfg.pooling == "att":
                rows.append((self.p(PFX + "attn.att_fc1.weight"), cfg.Qn, H, self.sh_a1, H, self.sh_a1T, QPAD))
            desc, start, tot = [], [0], 0
            for src, r, c, dst, ld, dstT, ldT in rows:
                desc.append([src.data_ptr(), r, c, dst.data_ptr(), ld, dstT.data_ptr() if dstT is not None else 0, ldT, 0])
                tot += ((r + 31) // 32) * ((c + 31) // 32)
                start.append(tot)
            return (torch.tensor(desc, dtype=torch.int64, device=dev), len(desc), tot,
                    torch.tensor(start, dtype=torch.int64, device=dev))

        self.desc_all = table(range(cfg.n_layers), True)
        self.desc_train = table(cfg.trainable_layers, True)

    def refresh_shadows(self, all_layers=False):
        d = self.desc_all if all_layers else self.desc_train
        self._c("tnr_refresh_shadows", d[0], d[1], d[2], d[3])
        if all_layers:
            self.fcache = None                 # every weight may have changed: the frozen-prefix cache is stale
            self.refresh_rel()

    def refresh_rel(self):
        """(A, Lr, Lr) additive rel-pos table of this engine's sequence length (depends on a frozen weight only)."""
        T.call("tnr_relpos_table", self.p(BERT + "rel_pos_bias.weight"), self.cfg.A, self.cfg.L, self.rel)
        self._rel_stale = False

    # ------------------------------------------------------------------ workspaces
    def _alloc_workspace(self, B):
        cfg, dev, bf = self.cfg, self.dev, self.tdt
        H, I, D, L = cfg.H, cfg.I, cfg.D, cfg.L
        N = B * (cfg.U + cfg.C)
        Mp = _rup(N * L + self.extra_rows, 128)
        self.B_alloc, self.N_alloc, self.Mp = B, N, Mp
        Nx = N + self.extra_seqs           # per-sequence partials / pooled vectors with room for a second pass's sequences
        if self.f16:
            # the losses are batch means, so activation gradients shrink as 1 / B: the scale grows with the batch in powers of two
            # beyond the B = 32 it was set for (B = 512 at the fixed scale: parameter gradients 1.8e-3 off the mean of their
            # B = 32 pieces; 1e-6 with it - a power-of-two batch then sees the very 16-bit values its pieces saw;
            # tools/scratch/big_batch.py, test_big_batch_equals_its_pieces).  B <= 32: unchanged.
            k = 0
            while (32 << k) < B and k < 5:
                k += 1
            self._gbase = LOSS_SCALE * (1 << k)
        z = lambda *s, dt=bf: torch.zeros(s, device=dev, dtype=dt)
        f = lambda *s: torch.zeros(s, device=dev, dtype=torch.float32)
        Lr = _rup(L, 32)
        self.Lr = Lr
        self.tok = torch.zeros((N, 2 * L), device=dev, dtype=torch.int64)
        self.nidx_buf = torch.zeros(N, device=dev, dtype=torch.int32)
        self.mask_add = f(N, Lr)
        self.rel = f(cfg.A, Lr, Lr)
        self.lse = f(N, cfg.A, Lr) if L > 32 else None        # long-sequence attention keeps the softmax statistics
        self.delta = f(N, cfg.A, Lr) if L > 32 else None
        self.x0 = z(Mp, H)
        n_keep = cfg.n_layers - self.lo
        mk = lambda: dict(qkv=z(Mp, 3 * H), ctx=z(Mp, H), h1pre=z(Mp, H), st1=f(Mp, 2), h1=z(Mp, H), u=z(Mp, I),
                          g=z(Mp, I), ypre=z(Mp, H), st2=f(Mp, 2), y=z(Mp, H),
                          lse=f(N, cfg.A, Lr) if L > 32 else None)
        self.act = [mk() for _ in range(n_keep)]          # resident activations of layers >= lo
        self.scr = mk() if self.lo > 0 else None           # scratch for frozen layers below lo
        self.scr_y = [z(Mp, H), z(Mp, H)] if self.lo > 0 else None
        self.e = f(Mp, QPAD)
        self.nv, self.alpha, self.den = f(Nx, H), f(N, Lr), f(N)
        Rt = N + B
        self.Rt = Rt
        if hasattr(self, "dyprem"):
            self.dyprem, self.dh1prem = z(Mp, H), z(Mp, H)
        self.S = f(Rt, D)                 # student rows: [B*U history | B*C candidate | B user]
        self.dS = f(Rt, D)
        self.Sv, self.dSv = f(N, D), f(N, D)      # vectors / gradients per DISTINCT news of the step (dedup.py)
        self.plan = None
        T_ = max(cfg.T, 1)
        self.X = f(T_, Rt, D)             # teacher rows, same layout
        self.Pm = f(T_, Rt, D)
        self.dP = f(T_, Rt, D)
        self.ones = torch.ones(Rt, device=dev)
        self.hidx = torch.arange(B * cfg.U, device=dev, dtype=torch.int32).view(B, cfg.U)
        self.cidx = (B * cfg.U + torch.arange(B * cfg.C, device=dev, dtype=torch.int32)).view(B, cfg.C)
        self.score, self.dscore = f(B, cfg.C), f(B, cfg.C)
        self.e_u, self.alpha_u, self.den_u = f(B, cfg.U, cfg.Qu), f(B, cfg.U), f(B)
        self.t_score = f(T_, B, cfg.C)
        self.e_t, self.alpha_t, self.den_t = f(T_, B, cfg.U, cfg.Qu), f(T_, B, cfg.U), f(T_, B)
        self.tw = f(B, T_)
        self.losses = f(4)
        self.kd_part = f(Rt)
        self.user_part = f(B, T.query("tnr_user_bwd_part_stride", D, cfg.Qu))     # [b1 | w2 | pad | b2] per impression
        self.hv_u, self.dhv_u, self.dpre_u = f(B * cfg.U, D), f(B * cfg.U, D), f(B * cfg.U, cfg.Qu)
        if cfg.nrms_heads:
            BU, BC = B * cfg.U, B * cfg.C
            mk = lambda nm: (f(nm, BU, D), f(nm, BU, 3 * D), f(nm, BU + BC, D))          # blended rows, q|k|v, [ctx | candidates]
            self.nr_s, self.nr_t = mk(1), mk(T_)
            self.nr_dctx, self.nr_dqkv, self.nr_dhv = f(BU, D), f(BU, 3 * D), f(BU, D)
            self.ones_mask = torch.ones((B, cfg.U), device=dev)
            self.hpos = torch.arange(BU, device=dev, dtype=torch.int32).view(B, cfg.U)
            self.cpos = (BU + torch.arange(max(BC, 1), device=dev, dtype=torch.int32))[:BC].view(B, cfg.C)
        # backward buffers
        self.dnv = f(Nx, H)
        self.dy, self.dy2 = z(Mp, H), z(Mp, H)
        self.dpre = z(Mp, QPAD)
        self.dw2p, self.db2p = f(Nx, cfg.Qn), f(Nx)
        self.dypre, self.dh1, self.dh1pre, self.dctx = z(Mp, H), z(Mp, H), z(Mp, H), z(Mp, H)
        self.du = z(Mp, I)
        self.dqkv = z(Mp, 3 * H)
        # partial sums of the bias / LayerNorm gradients: a set per trainable layer, so that ONE batched reduction at the end of the
        # backward can sum them all (they are a few tens of MB per layer)
        self.lpart = {l: dict(ln_part=f(T.query("tnr_ln_bwd_part_elems", Mp, H)), ln_part1=f(T.query("tnr_ln_bwd_part_elems", Mp, H)),
                              gcs_part=f(T.query("tnr_gemm_colsum_rows", Mp), I), qkvb_part=f(Nx, 3 * H), cs_tmp=f(I), cs_tmp2=f(3 * H))
                      for l in cfg.trainable_layers}
        self.red = {}                                                      # gradient bucket -> _ReduceBatch
        self.cs_part = f(max(T.query("tnr_colsum_part_elems", Mp, 3 * H if L > 32 else QPAD),
                             T.query("tnr_colsum_part_elems", 128, I),
                             T.query("tnr_colsum_part_elems", max(B * cfg.U, 1), 3 * D),
                             T_ * T.query("tnr_colsum_part_elems", Rt, D)))
        self.db1p = f(Nx, QPAD)
        # chunked pooling kernels for few, long sequences (stage-1 bodies): their workspace
        self.ap_ws = f(T.query("tnr_attpool_long_ws_elems", N, L, H, cfg.Qn, QPAD)) if (L > AP_LONG and cfg.pooling == "att") else None
        self.epre_u = f(B * cfg.U, cfg.Qu)
        self.epad_buf = {1: f(1, cfg.Qu), T_: f(T_, cfg.Qu)}      # student / teachers (these run on different streams)
        self.epre_t = f(T_, B * cfg.U, cfg.Qu)
        self.KS = 8                                                        # split-K of the long-K small GEMMs
        # x 2: the K-split members of a grouped launch get disjoint slices (_sgemm_group)
        self.sg_part = f(2 * self.KS * max(T_ * D * D, D * H, 3 * D * D, T_ * Rt * D, N * max(H, D), T_ * B * cfg.U * max(cfg.Qu, 3 * D)))
        self.sg_part2 = f(self.KS * max(cfg.Qu * D, 3 * D * D if cfg.nrms_heads else 1))       # second split problem of a grouped launch
        # weight-gradient slabs: the four gradients of a layer are in flight together (_wgrad_flush), each with slabs of its own
        self.ws = f(max(sum(_rup(self._wgrad_splits(n_, k_)[1], 64) for n_, k_ in ((3 * H, H), (H, H), (I, H), (H, I))),
                        self._wgrad_splits(QPAD, H)[1]))
        self._rel_stale = True       # filled by the first encode() (no kernel launch at construction time)

    WGRAD_UNITS = 1     # work units (split, tile) per workgroup of the queue-fed weight-gradient kernel.  1 = one full round
                        # (fastest alone).  Beside CUs held by an overlapped all-reduce a late workgroup costs a whole unit:
                        # x1.65 at 1, x1.25 at 2, x1.1-1.15 at 4 -- but every doubling adds a round of fp32 slab traffic
                        # (alone: +10 % at 2, +24 % at 4 for the 3072 x 768 gradient; tools/cu_contention.py)

    merge_reductions = True   # False: the partial-sum reductions per gradient bucket even without a bucket hook (tools/step_ab.py)
    group_wgrad = False  # True: a layer's weight gradients in one persistent launch (tnr_gemm_tn_wgrad_group; the same bits).  Measured
                         # on one GPU: step +6.4 %, the four gradients 848 -> 1 088 us (EXPERIMENTS.md section 4 item 22) - one launch each,
                         # one unit per workgroup, all units equal and in lockstep, is the better schedule; kept as a switch for the
                         # data-parallel case, where a workgroup held off its CU by a collective costs a quarter of what it costs now
    _wg = None           # the collected weight gradients while a layer's backward runs
    _wg_defer = None     # ... while a backward runs step by step beside another engine's (backward_encoder_steps)

    @staticmethod
    def _wgrad_splits(N, K, units=None):
        """-> (splits over M, workspace elems): `units` (default Engine.WGRAD_UNITS) rounds of workgroups on the 256 CUs (one
        256x256 output tile x split each); more splits only add fp32 slab traffic (N*K*4 B written and re-read per split)."""
        tiles = (N // 256) * (K // 256) if (N % 256 == 0 and K % 256 == 0) else (N // 128) * (K // 128)
        splits = max(1, min(64, (units or Engine.WGRAD_UNITS) * 256 // max(tiles, 1)))
        return splits, T.query("tnr_gemm_tn_ws_elems", N, K, splits)

    # ------------------------------------------------------------------ kernel wrappers
    def _gemm(self, a, w, c, M, bias=None, res=None, aux=None, flags=0, colsum=None, drop=None):
        N, K = w.shape
        args = (a, a.stride(0), w, w.stride(0), c, c.stride(0), M, N, K, bias, res,
                res.stride(0) if res is not None else 0, aux, aux.stride(0) if aux is not None else 0, flags, colsum)
        if drop is None:
            self._c("tnr_gemm_nt_ex", *args)
        else:
            self._c("tnr_gemm_nt_do", *args, drop)

    def _wgrad(self, dy, x, dw, M, acc=0):
        N, K = dw.shape
        if self._wg is not None:           # inside a layer's backward: collected, launched together by _wgrad_flush
            self._wg.append((dy, x, dw, M, N, K, acc))
            return
        if self._wg_defer is not None:     # stage 1: collected for a launch shared with the other pass's (Stage1Engine.backward)
            self._wg_defer.append((dy, x, dw, M, N, K, acc))
            return
        self._c("tnr_gemm_tn_wgrad_ex", dy, dy.stride(0), x, x.stride(0), dw, dw.stride(0), M, N, K, self.ws,
               self._wgrad_splits(N, K)[0], acc, self.ginv)

    def _wgrad_flush(self):
        """The weight gradients collected since the last flush in ONE persistent launch (+ one slab-sum launch): their (split, tile)
        units share a queue, so the workgroups that finish one gradient's units go on with the next one's instead of idling to the
        launch boundary, and a workgroup kept off its CU by another stream's kernel costs a unit of up to four rounds' worth
        instead of a whole round.  Each gradient is computed exactly as by its own launch (slabs of its own in self.ws)."""
        pend, self._wg = self._wg, []
        if not pend:
            return
        probs, off = [], 0
        for dy, x, dw, M, N, K, acc in pend:
            splits, elems = self._wgrad_splits(N, K)
            probs.append(dict(dY=dy, lddy=dy.stride(0), X=x, ldx=x.stride(0), dW=dw, lddw=dw.stride(0), M=M, N=N, K=K,
                              ws=self.ws[off:off + elems], splits=splits, accumulate=acc, out_scale=self.ginv))
            off += _rup(elems, 64)
        assert off <= self.ws.numel()
        for i in range(0, len(probs), 4):
            T.wgrad_group(probs[i:i + 4], f16=self.f16)

    def _colsum(self, x, out, M, dtype=T.BF16):   # dtype BF16 = "the 16-bit type of the build"
        self._c("tnr_colsum", x, x.stride(0), dtype, M, x.shape[1], out, self.cs_part, 0)

    def _sgemm(self, A, a_rs, a_cs, sA, Bm, b_rs, b_cs, sB, C, ldc, sC, bias, sBias, M, N, K, batch=1, ksplit=None, alpha=1.0,
               beta=0.0):
        """ksplit None: K is split into slices of >= 256 (a workgroup's time on the fp32 MFMA is ~K * 32 cycles whatever the
        tile count, and these GEMMs have few tiles; slices of 128 cost a split-reduce launch for the K = 256 GEMMs and were 0.3 %
        slower on the step, `AB=sg_kdiv:128:256 tools/step_ab.py`).  The split depends on K ALONE, never on the row count: a news vector must
        come out with the same bits whatever batch it is encoded in (in-batch de-duplication, frozen-layer cache)."""
        if ksplit is None:
            ksplit = max(1, min(K // getattr(self, "sg_kdiv", 256), 8))
            assert ksplit * batch * M * N <= self.sg_part.numel(), "fp32 GEMM workspace too small for its K split"
        T.call("tnr_sgemm", A, a_rs, a_cs, sA, None, Bm, b_rs, b_cs, sB, C, ldc, sC, bias, sBias, M, N, K, batch, alpha, beta,
               ksplit, self.sg_part if ksplit > 1 else None)

    def _sgemm_problem(self, A, a_rs, a_cs, sA, Bm, b_rs, b_cs, sB, C, ldc, sC, bias, sBias, M, N, K, batch=1, ksplit=None, alpha=1.0,
                       beta=0.0, part=None):
        """The arguments of _sgemm as one member of a grouped launch (T.sgemm_group: independent small GEMMs in one launch, each
        computed exactly as its own tnr_sgemm call would).  A split problem needs a partial buffer of its own (`part`)."""
        if ksplit is None:
            ksplit = max(1, min(K // getattr(self, "sg_kdiv", 256), 8))
        if ksplit > 1 and part is not None:
            assert ksplit * batch * M * N <= part.numel(), "fp32 GEMM workspace too small for its K split"
        return dict(A=A, a_rs=a_rs, a_cs=a_cs, sA=sA, B=Bm, b_rs=b_rs, b_cs=b_cs, sB=sB, C=C, ldc=ldc, sC=sC, bias=bias, sBias=sBias,
                    M=M, N=N, K=K, batch=batch, alpha=alpha, beta=beta, ksplit=ksplit, part=part if ksplit > 1 else None)

    def _sgemm_group(self, problems):
        """Independent fp32 GEMMs in one launch (Engine.group_sgemm = False: one tnr_sgemm each, the same bits; tools/step_ab.py).
        The members run concurrently, so every K-split member gets a DISJOINT slice of the partial buffer (tnr_sgemm_group:
        'not shared inside a group'); a member that brought its own buffer keeps it."""
        off = 0
        for q in problems:
            if q["ksplit"] > 1 and q["part"] is None:
                need = q["ksplit"] * q["batch"] * q["M"] * q["N"]
                q["part"] = self.sg_part[off:off + need]
                off += _rup(need, 64)
        assert off <= self.sg_part.numel(), "fp32 GEMM workspace too small for the K splits of a grouped launch"
        parts = [q["part"].data_ptr() for q in problems if q["part"] is not None]
        assert len(parts) == len(set(parts)), "two members of a grouped fp32 GEMM share a partial buffer"
        if getattr(self, "group_sgemm", True):
            T.sgemm_group(problems)
            return
        for q in problems:
            T.call("tnr_sgemm", q["A"], q["a_rs"], q["a_cs"], q["sA"], None, q["B"], q["b_rs"], q["b_cs"], q["sB"], q["C"], q["ldc"], q["sC"],
                   q["bias"], q["sBias"], q["M"], q["N"], q["K"], q["batch"], q["alpha"], q["beta"], q["ksplit"], q["part"])

    # ------------------------------------------------------------------ forward
    def build_frozen_cache(self, news_combined):
        """The layers below the first trainable one never change during training (run.py:101-112), so for a resident
        news table their output is a function of the news id alone: compute it ONCE for every row -- hidden states
        entering layer `lo` (n+1, L*H) 16-bit, 2.4 GB for 51 k titles, plus the additive masks -- and let encode() gather
        rows instead of re-running the embedding and the frozen layers every step.  Bit-identical to recomputing
        (rows are processed independently); the reference recomputes, bench.py's headline does too."""
        self.fcache = None
        if self.lo == 0:
            return False
        n, L, H = news_combined.shape[0], self.cfg.L, self.cfg.H
        fx = torch.empty((n, L * H), device=self.dev, dtype=self.tdt)
        fm = torch.empty((n, self.Lr), device=self.dev, dtype=torch.float32)
        cap = self.N_alloc
        for s0 in range(0, n, cap):
            cnt = min(cap, n - s0)
            idx = torch.arange(s0, s0 + cnt, device=self.dev, dtype=torch.int32)
            x = self.encode(news_combined, cnt, nidx=idx, stop_at=self.lo, train=False)
            fx[s0:s0 + cnt].copy_(x[:cnt * L].view(cnt, L * H))
            fm[s0:s0 + cnt].copy_(self.mask_add[:cnt])
        self.fcache = (fx.view(torch.float32), fm, news_combined.data_ptr(), n)
        return True

    def _pos_ids(self, tok):
        """RoBERTa position ids of a token table (rows [ids | mask], any integer dtype) -> (rows, L) int32; the resident table's
        are computed once (cached by its address).  Integer index preparation, like the torch.cat of the news indices."""
        key = (tok.data_ptr(), tuple(tok.shape))
        c = getattr(self, "_pos_cache", None)
        if c is not None and c[0] == key:
            return c[1]
        L, pad = self.cfg.L, self.cfg.pos_pad_id
        ne = (tok[:, :L] != pad)
        pid = (torch.cumsum(ne.to(torch.int32), 1) * ne + pad).to(torch.int32).contiguous()
        if tok.shape[0] > 4 * self.N_alloc:          # a resident table, not a per-step batch
            self._pos_cache = (key, pid)
        return pid

    # pieces of encode() / backward_encoder_steps() that depend on the sequence length, with their per-token operands explicit
    # (stage 1's joint passes run them per pass on row ranges of buffers both passes share, stage1.py)
    def _embed_fwd(self, tok, n_seq, nidx, x0):
        cfg, g, ds = self.cfg, self.p, self._dsite
        L, H = cfg.L, cfg.H
        emb = (g(BERT + "embeddings.word_embeddings.weight"), g(BERT + "embeddings.position_embeddings.weight"),
               g(BERT + "embeddings.token_type_embeddings.weight"), g(BERT + "embeddings.LayerNorm.weight"),
               g(BERT + "embeddings.LayerNorm.bias"), cfg.ln_eps, x0, self.mask_add)
        de = ds(T.DROP_EMB, 0)
        pid = self._pos_ids(tok if nidx is not None else tok[:n_seq]) if cfg.pos_pad_id is not None else None
        if nidx is None:
            self._c("tnr_embed_ln_fwd_do", tok, n_seq, L, H, *emb, de, pid) if (de or pid is not None) else \
                self._c("tnr_embed_ln_fwd", tok, n_seq, L, H, *emb)
        else:
            self._c("tnr_embed_ln_fwd_indexed_do", tok, nidx, n_seq, L, H, *emb, de, pid) if (de or pid is not None) else \
                self._c("tnr_embed_ln_fwd_indexed", tok, nidx, n_seq, L, H, *emb)

    def _attn_fwd(self, qkv, ctx, lse, n_seq, dp):
        cfg = self.cfg
        if cfg.L <= 32:
            if dp:
                self._c("tnr_attn_l32_fwd_do", qkv, self.mask_add, self.rel, ctx, n_seq, cfg.L, cfg.A, dp)
            else:
                self._c("tnr_attn_l32_fwd", qkv, self.mask_add, self.rel, ctx, n_seq, cfg.L, cfg.A)
        else:
            largs = (qkv, self.mask_add, self.rel, ctx, lse, n_seq, cfg.L, cfg.A)
            self._c("tnr_attn_long_fwd_do", *largs, dp) if dp else self._c("tnr_attn_long_fwd", *largs)

    def _attpool_fwd(self, x, e, nv, n_seq):
        cfg, g = self.cfg, self.p
        if cfg.L > AP_LONG:                  # few, long sequences (stage-1 bodies): the chunked kernels fill the chip
            self._c("tnr_attpool_fwd_long", x, e, QPAD, g(PFX + "attn.att_fc2.weight"), g(PFX + "attn.att_fc2.bias"), cfg.Qn,
                    nv, self.alpha, self.den, self.ap_ws, n_seq, cfg.L, cfg.H)
        else:
            self._c("tnr_attpool_fwd", x, e, QPAD, g(PFX + "attn.att_fc2.weight"), g(PFX + "attn.att_fc2.bias"), cfg.Qn,
                    nv, self.alpha, self.den, n_seq, cfg.L, cfg.H)

    def _attpool_bwd(self, y, e, dnv, dy2, dpre, dw2p, db2p, db1p, n_seq):
        cfg, g = self.cfg, self.p
        if cfg.L > AP_LONG:
            self._c("tnr_attpool_bwd_long", y, e, QPAD, g(PFX + "attn.att_fc2.weight"), cfg.Qn, dnv, self.alpha,
                    dy2, dpre, QPAD, dw2p, db2p, db1p, self.ap_ws, n_seq, cfg.L, cfg.H)
        else:
            self._c("tnr_attpool_bwd", y, e, QPAD, g(PFX + "attn.att_fc2.weight"), cfg.Qn, dnv, self.alpha, self.den,
                    dy2, dpre, QPAD, dw2p, db2p, db1p, n_seq, cfg.L, cfg.H)

    def _attn_bwd(self, qkv, ctx, lse, dctx, dqkv, qkvb_part, n_seq, dPb):
        """-> True if the q/k/v bias partial sums came out of the kernel (L <= 32), False if the caller sums dqkv's columns."""
        cfg = self.cfg
        if cfg.L <= 32:
            bargs = (qkv, self.mask_add, self.rel, dctx, dqkv, qkvb_part, n_seq, cfg.L, cfg.A)
            self._c("tnr_attn_l32_bwd_do", *bargs, dPb) if dPb else self._c("tnr_attn_l32_bwd", *bargs)
            return True
        bargs = (qkv, self.mask_add, self.rel, ctx, dctx, lse, self.delta, dqkv, n_seq, cfg.L, cfg.A)
        self._c("tnr_attn_long_bwd_do", *bargs, dPb) if dPb else self._c("tnr_attn_long_bwd", *bargs)
        return False

    def encode(self, tok, n_seq, nidx=None, out=None, stop_at=None, train=True, extra=None):
        """NewsEncoder.forward model_bert.py:119-137 -> news vectors S[:n_seq] (fp32).
        tok (n_seq, 2L) int64 on device, or (nidx given) tok = resident news_combined (n+1, 2L) int32 and
        nidx (n_seq,) int32 news indices.  stop_at = l: return the hidden states entering layer l instead."""
        cfg = self.cfg
        H, L = cfg.H, cfg.L
        M = n_seq * L
        g = self.p
        if self._rel_stale:
            self.refresh_rel()
        self.drop_cur = None
        if self.drop is not None and train:
            self.drop_cur = dict(self.drop, call=2 * self.drop_calls + self.drop["pass_id"])
            self.drop_calls += 1
        ds = self._dsite
        fc = getattr(self, "fcache", None) if self.drop_cur is None else None      # a cached prefix has no fresh masks
        first = 0
        if fc is not None and nidx is not None and stop_at is None and tok.data_ptr() == fc[2]:
            # frozen prefix from the per-news cache (build_frozen_cache): two row gathers replace embedding + lo layers
            T.call("tnr_gather_rows", fc[0], fc[3], nidx, n_seq, fc[0].shape[1], 1, self.x0.view(torch.float32), n_seq, 0)
            T.call("tnr_gather_rows", fc[1], fc[3], nidx, n_seq, self.Lr, 1, self.mask_add, n_seq, 0)
            first = self.lo
        else:
            self._embed_fwd(tok, n_seq, nidx, self.x0)
        x = self.x0
        self.x_in = {}
        for l in range(first, cfg.n_layers):
            if stop_at is not None and l == stop_at:
                return x
            names = layer_param_order(l)
            sh = self.sh[l]
            kept = l >= self.lo
            a = self.act[l - self.lo] if kept else self.scr
            y = a["y"] if kept else self.scr_y[l & 1]
            bqkv = self._view(names[3], 3 * H, (3 * H,))
            self.x_in[l] = x
            self._gemm(x, sh["qkv"], a["qkv"], M, bias=bqkv, flags=T.EPI_BIAS)
            self._attn_fwd(a["qkv"], a["ctx"], a["lse"] if kept else self.lse, n_seq, ds(T.DROP_PROB, l))
            self._gemm(a["ctx"], sh["o"], a["h1pre"], M, bias=g(names[7]), res=x, flags=T.EPI_BIAS | T.EPI_RES,
                       drop=ds(T.DROP_ATTN_OUT, l))
            self._c("tnr_ln_fwd", a["h1pre"], g(names[8]), g(names[9]), cfg.ln_eps, a["h1"], a["st1"], M, H)
            fl = T.EPI_BIAS | T.EPI_GELU | (T.EPI_AUXOUT if kept else 0)
            self._gemm(a["h1"], sh["w1"], a["g"], M, bias=g(names[11]), aux=a["u"] if kept else None, flags=fl)
            self._gemm(a["g"], sh["w2"], a["ypre"], M, bias=g(names[13]), res=a["h1"], flags=T.EPI_BIAS | T.EPI_RES,
                       drop=ds(T.DROP_FFN_OUT, l))
            self._c("tnr_ln_fwd", a["ypre"], g(names[14]), g(names[15]), cfg.ln_eps, y, a["st2"], M, H)
            x = y
        self.y_last = x
        # pooling (model_bert.py:130-135: AttentionPooling without mask | token 0 | mean) + dense (:136)
        if cfg.pooling == "att":
            self._gemm(x, self.sh_a1, self.e, M, bias=self.b_a1, flags=T.EPI_BIAS | T.EPI_TANH | T.EPI_OUTF32)
            self._attpool_fwd(x, self.e, self.nv, n_seq)
        else:
            self._c("tnr_pool_fwd", x, self.nv, n_seq, L, H, int(cfg.pooling == "mean"))
        wd = g(PFX + "dense.weight")
        dst = self.S if out is None else out
        dense = (self.nv, H, 1, 0, wd, H, 1, 0, dst, cfg.D, 0, g(PFX + "dense.bias"), 0, n_seq, cfg.D, H)
        if extra:      # independent fp32 GEMMs the caller had pending (the teachers' projection): one launch with the dense layer
            self._sgemm_group([self._sgemm_problem(*dense)] + list(extra))
        else:
            self._sgemm(*dense)
        return dst[:n_seq]

    # ------------------------------------------------------------------ forward-only paths
    @torch.no_grad()
    def encode_news(self, news_combined):
        """news_scoring of run.py:276-287 / :438-444: every row of the resident token table (n+1, 2L) int32
        through NewsEncoder.forward, in passes of the workspace's sequence capacity.  -> (n+1, D) fp32."""
        n = news_combined.shape[0]
        cap = self.N_alloc
        out = torch.empty((n, self.cfg.D), device=self.dev, dtype=torch.float32)
        for s0 in range(0, n, cap):
            cnt = min(cap, n - s0)
            idx = torch.arange(s0, s0 + cnt, device=self.dev, dtype=torch.int32)
            out[s0:s0 + cnt].copy_(self.encode(news_combined, cnt, nidx=idx, train=False))
        return out

    @torch.no_grad()
    def user_vectors(self, news_scoring, hist_idx, history_mask):
        """student.user_encoder(log_vecs, log_mask) of run.py:343 on rows of news_scoring gathered by index.
        hist_idx (B,U) int32, mask (B,U) -> (B,D) fp32 (B <= the engine's batch)."""
        cfg = self.cfg
        B, U, D, Qu = hist_idx.shape[0], cfg.U, cfg.D, cfg.Qu
        assert B <= self.B_alloc and hist_idx.shape[1] == U
        g, ue = self.p, "student.user_encoder."
        rows = self.X[0, :B * U]                                   # scratch: gathered history rows
        T.call("tnr_gather_rows", news_scoring, news_scoring.shape[0], hist_idx.reshape(-1).contiguous(), B * U, D, 1, rows,
               B * U, 0)
        hidx = self.hidx[:B]
        mask = history_mask.to(torch.float32).contiguous()
        self._user_forward(rows, B * U, 1, self._user_params(ue, 1), hidx, hidx, mask, self.epre_u, self.dS, B * D, self.score,
                           self.e_u, self.alpha_u, self.den_u, B, 0, getattr(self, "nr_s", None))
        return self.dS[:B].clone()

    def _prepare(self, B):
        # buffers are sized for the exact batch: row strides of the stacked (T, Rt, D) arrays and the
        # zero rows the wgrad kernels rely on (rows [M, roundup(M,64))) both depend on it.  Only the last,
        # short batch of an epoch (streaming.py:76, no drop_remainder) ever changes B.
        if B != self.B_alloc:
            self._alloc_workspace(B)

    def forward_indexed(self, news_combined, hist_idx, history_mask, cand_idx, label, teacher_tables, plan=None):
        """The same step fed the way the resident-table loader feeds it: news_combined (n+1, 2L) int32 and
        teacher_tables (T, n+1, D) fp32 stay in HBM; per step only hist_idx (B,U), cand_idx (B,C) int32 news
        indices, the mask and the labels arrive (dataloader.py:129-149 at index level)."""
        return self.forward(None, history_mask, None, label, teacher_tables=teacher_tables, t_hidx=hist_idx,
                            t_cidx=cand_idx, news_combined=news_combined, plan=plan)

    def forward(self, history, history_mask, candidate, label, teacher_hist=None, teacher_cand=None,
                teacher_tables=None, t_hidx=None, t_cidx=None, news_combined=None, plan=None):
        """Model.forward model_bert.py:262-305.  Teacher embeddings either as the reference's lists of
        (B,U,D)/(B,C,D) tensors, or as resident tables (T,R,D) + int32 row ids (B,U)/(B,C).
        Returns the device tensor [distill, target, emb, -] and student_score (B,C)."""
        cfg = self.cfg
        B = history_mask.shape[0]
        U, C, L, D, T_ = cfg.U, cfg.C, cfg.L, cfg.D, cfg.T
        self._prepare(B)
        N = B * (U + C)
        Rt = N + B
        self.cur = (B, N, Rt)
        self.mask = history_mask.to(torch.float32).contiguous()
        self.label = label.to(torch.int64).contiguous()
        idx = None
        if news_combined is not None or teacher_tables is not None:
            if (t_hidx.dtype == torch.int32 and t_cidx.dtype == torch.int32 and t_hidx.is_contiguous() and t_cidx.is_contiguous()
                    and t_hidx.device == self.dev):
                idx = self.nidx_buf[:N]                  # no ATen kernel inside the step
                T.call("tnr_concat_i32", t_hidx, B * U, t_cidx, B * C, idx)
            else:
                idx = torch.cat([t_hidx.reshape(-1), t_cidx.reshape(-1)]).to(device=self.dev, dtype=torch.int32)
        self.nidx = idx
        # teacher side (frozen user encoders over the teachers' rows, projection by the transform matrices): independent of
        # the student's encoder pass; issued first (optionally on a second stream, TNR_TEACHER_STREAM=1)
        hidx, cidx = self._idx(B)
        main = torch.cuda.current_stream(self.dev) if self.dev.type == "cuda" else None
        if T_ > 0:
            side = self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if teacher_tables is not None:
                    T.call("tnr_gather_rows", teacher_tables, teacher_tables.shape[1], idx, N, D, T_, self.X, self.X.shape[1], 0)
                else:
                    for i in range(T_):
                        self.X[i, :B * U].copy_(teacher_hist[i].reshape(B * U, D))
                        self.X[i, B * U:N].copy_(teacher_cand[i].reshape(B * C, D))
                self._user_forward(self.X, self.X.shape[1], T_, self._user_params("teachers.0.", T_), hidx, cidx, self.mask,
                                   self.epre_t, self.X[0, N:], self.X.stride(0), self.t_score, self.e_t, self.alpha_t,
                                   self.den_t, B, C, getattr(self, "nr_t", None))
                Wt = self._view("transform_matrix.0.weight", T_ * D * D, (T_, D, D))
                bt = self._view("transform_matrix.0.bias", T_ * D, (T_, D))
                proj = (self.X, D, 1, self.X.stride(0), Wt, D, 1, D * D, self.Pm, D, self.Pm.stride(0), bt, D, Rt, D, D)
                if side == main:       # the projection rides in the launch of the news encoder's dense layer (encode(extra=...))
                    extra = [self._sgemm_problem(*proj, batch=T_)]
                else:
                    self._sgemm(*proj, batch=T_)
        if T_ == 0 or side != main:
            extra = None
        if news_combined is not None:
            self.plan = plan
            if plan is None:
                self.encode(news_combined, N, nidx=self.nidx, extra=extra)
            else:
                # encode each distinct news once, expand to the slots (dedup.py); everything downstream is unchanged
                assert plan.n_slots == N and plan.n_enc <= self.N_alloc
                self.encode(news_combined, plan.n_enc, nidx=plan.uniq, out=self.Sv, extra=extra)
                T.call("tnr_gather_rows", self.Sv, plan.n_enc, plan.inv, N, D, 1, self.S, self.S.shape[0], 0)
        else:
            assert history.shape[1:] == (U, 2 * L) and candidate.shape[1:] == (C, 2 * L)
            self.plan = None
            tok = self.tok[:N]
            tok[:B * U].copy_(history.reshape(B * U, 2 * L))
            tok[B * U:].copy_(candidate.reshape(B * C, 2 * L))
            self.encode(tok, N, extra=extra)
        S = self.S[:Rt]
        g = self.p
        Qu = cfg.Qu
        self._user_forward(S, Rt, 1, self._user_params("student.user_encoder.", 1), hidx, cidx, self.mask, self.epre_u, S[N:],
                           B * D, self.score, self.e_u, self.alpha_u, self.den_u, B, C, getattr(self, "nr_s", None))
        if T_ > 0:
            main.wait_stream(side)
        T.call("tnr_kd_score_loss", self.score, self.t_score if T_ else None, self.label, cfg.temperature, cfg.coef,
               self.tw if T_ else None, self.dscore, self.losses, B, C, T_)
        if T_ > 0:
            T.call("tnr_kd_embed_loss", S, self.Pm, self.tw, self.losses[2:], self.dS, self.dP, self.kd_part, B, U, C, D, T_)
        else:
            self.dS[:Rt].zero_()
            self.losses[2:3].zero_()
        return self.losses, self.score[:B]

    def _user_params(self, first, nm):
        """Stacked parameter views of `nm` user encoders whose first member has prefix `first`."""
        cfg = self.cfg
        D, Qu = cfg.D, cfg.Qu
        v = lambda suffix, per, shape: self._view(first + suffix, nm * per, (nm,) + shape)
        p = dict(pad=v("pad_doc", D, (D,)), w1=v("attn.att_fc1.weight", Qu * D, (Qu, D)), b1=v("attn.att_fc1.bias", Qu, (Qu,)),
                 w2=v("attn.att_fc2.weight", Qu, (Qu,)), b2=v("attn.att_fc2.bias", 1, ()))
        if cfg.nrms_heads:
            p["wqkv"] = v(NRMS_W[0], 3 * D * D, (3 * D, D))
            p["bqkv"] = v(NRMS_B[0], 3 * D, (3 * D,))
        return p

    def _user_forward(self, vec, R, nm, p, hidx, cidx, mask, epre, user, user_stride, score, e, alpha, den, B, C, nr):
        """UserEncoder.forward (model_bert.py:155-176) + scorer bmm for `nm` stacked encoders over the row tables
        vec (nm, R, D): history rows hidx (B,U), candidate rows cidx (B,C)."""
        cfg = self.cfg
        U, D, Qu, ulm = cfg.U, cfg.D, cfg.Qu, int(cfg.user_log_mask)
        if not cfg.nrms_heads:
            if getattr(self, "fused_user_fwd", True) and D % 8 == 0 and 4 * (64 * (D + 4) + U * Qu + 64 + D) <= 160 * 1024:
                # fc1 inside the kernel: one launch per pass (was: batched fp32 GEMM + split reduce + fc1(pad) pair + this kernel)
                T.call("tnr_user_score_fwd", vec, R, hidx, cidx, mask, p["pad"], p["w1"], p["b1"], p["w2"], p["b2"], ulm, None, None,
                       user, user_stride, score, e, alpha, den, nm, B, U, C, D, Qu)
                return
            sv = vec.stride(0) if vec.dim() == 3 else 0
            self._sgemm(vec, D, 1, sv, p["w1"], D, 1, Qu * D, epre, Qu, B * U * Qu, p["b1"], Qu, B * U, Qu, D, batch=nm)
            epad = None
            if not ulm:        # fc1(pad_doc) once per model instead of once per (impression, model) workgroup
                epad = self.epad_buf[nm]
                self._sgemm(p["pad"], D, 1, D, p["w1"], D, 1, Qu * D, epad, Qu, Qu, p["b1"], Qu, 1, Qu, D, batch=nm)
            T.call("tnr_user_score_fwd", vec, R, hidx, cidx, mask, p["pad"], p["w1"], p["b1"], p["w2"], p["b2"], ulm, epre, epad,
                   user, user_stride, score, e, alpha, den, nm, B, U, C, D, Qu)
            return
        # NRMS (:162-164, :171-173): blend -> self-attention over the clicked news -> additive pooling of its output
        hv, qkv, ctx = nr
        BU = B * U
        T.call("tnr_user_blend_fwd", vec, R, hidx, mask, p["pad"], ulm, hv, nm, B, U, D)
        self._sgemm(hv, D, 1, hv.stride(0), p["wqkv"], D, 1, 3 * D * D, qkv, 3 * D, qkv.stride(0), p["bqkv"], 3 * D, BU, 3 * D, D,
                    batch=nm)
        T.call("tnr_nrms_attn_fwd", qkv, mask, ulm, ctx, ctx.shape[1], nm, B, U, cfg.nrms_heads)
        if C:
            T.call("tnr_gather_rows", vec, R, cidx.reshape(-1), B * C, D, nm, ctx, ctx.shape[1], BU)
        self._sgemm(ctx, D, 1, ctx.stride(0), p["w1"], D, 1, Qu * D, epre, Qu, BU * Qu, p["b1"], Qu, BU, Qu, D, batch=nm)
        # the pooling sees the attention output as it is: no blend (done above), mask only under user_log_mask
        T.call("tnr_user_score_fwd", ctx, ctx.shape[1], self.hpos[:B], self.cpos[:B], mask if ulm else self.ones_mask[:B], p["pad"],
               p["w1"], p["b1"], p["w2"], p["b2"], 1, epre, None, user, user_stride, score, e, alpha, den, nm, B, U, C, D, Qu)

    def _side_stream(self):
        # measured on MI355X (A/B on one box): no gain from a real second stream for the teacher side, so it shares
        # the main one (Engine.teacher_stream = a torch.cuda.Stream to opt in)
        side = getattr(self, "teacher_stream", None)
        return side if side is not None else torch.cuda.current_stream(self.dev)

    def _idx(self, B):
        assert B == self.B_alloc
        return self.hidx, self.cidx

    def total_loss(self):
        """distill + coef*target + emb  (model_bert.py:305) as a device scalar."""
        l = self.losses
        return l[0] + self.cfg.coef * l[1] + l[2]

    # ------------------------------------------------------------------ backward
    def _red_check(self):
        """The batched reductions replay a recorded job table that carries 1 / (loss scale) in its descriptors: when the dynamic
        scale has moved since they were recorded (an fp16 overflow, a growth step) they are recorded afresh."""
        if getattr(self, "_red_ginv", None) != self.ginv:
            self.red.clear()
            self._red_ginv = self.ginv

    def backward(self, after_bucket=None):
        """d total_loss / d trainable parameters -> self.flat_g.  after_bucket(i) is called when gradient
        bucket i (0 = heads, then one per layer from the top) is complete (dist.py overlaps its all-reduce)."""
        self._red_check()
        cfg = self.cfg
        B, N, Rt = self.cur
        U, C, D, T_ = cfg.U, cfg.C, cfg.D, cfg.T
        g = self.p
        S, dS = self.S[:Rt], self.dS
        hidx, cidx = self._idx(B)
        rbh = self.red.setdefault(("heads", 0, N if self.plan is None else self.plan.n_enc, after_bucket is None and self.merge_reductions),
                                  _ReduceBatch(self.dev))   # backward_encoder's key
        pend = []                 # fp32 GEMMs that depend on nothing computed below: launched together with the user encoder's
        if T_ > 0:
            self._transform_grads(Rt, rbh, pend)
        # scorer + user encoder
        T.call("tnr_score_bwd", S, cidx, S[N:], self.dscore, dS, dS[N:], B, C, D)
        ue = "student.user_encoder."
        Qu, ulm = cfg.Qu, int(cfg.user_log_mask)
        w1u = g(ue + "attn.att_fc1.weight")
        nrms = bool(cfg.nrms_heads)
        # pooling backward: over the history rows themselves, or (NRMS) over the self-attention output
        pv, ph, pm, pu, pd = (S, hidx, self.mask, ulm, dS) if not nrms else \
            (self.nr_s[2][0], self.hpos, self.mask if ulm else self.ones_mask, 1, self.nr_dctx)
        T.call("tnr_user_bwd_pre", pv, ph, pm, g(ue + "pad_doc"), g(ue + "attn.att_fc2.weight"), pu, dS[N:], self.e_u,
               self.alpha_u, self.hv_u, self.dpre_u, self.user_part, B, U, D, Qu)
        # dW1 = dpre^T hv (straight into the gradient) ; dhv = dpre W1      -- fp32 MFMA GEMMs
        self._sgemm_group(pend + [
            self._sgemm_problem(self.dpre_u, 1, Qu, 0, self.hv_u, 1, D, 0, self.grads[ue + "attn.att_fc1.weight"], D, 0, None, 0,
                                Qu, D, B * U, ksplit=self.KS, part=self.sg_part2),
            self._sgemm_problem(self.dpre_u, Qu, 1, 0, w1u, 1, D, 0, self.dhv_u, D, 0, None, 0, B * U, D, Qu)])
        if nrms:
            pd.zero_()
        T.call("tnr_user_bwd_post", self.dhv_u, self.alpha_u, dS[N:], pm, ph, pu, pd, self.user_part, B, U, D, Qu)
        if nrms:
            # self-attention backward (model_bert.py:62-100), then the blend in front of it
            hv, qkv, _ = self.nr_s
            BU = B * U
            T.call("tnr_nrms_attn_bwd", qkv, self.mask, ulm, self.nr_dctx, self.nr_dqkv, B, U, cfg.nrms_heads)
            wqkv = self._view(ue + NRMS_W[0], 3 * D * D, (3 * D, D))
            self._sgemm(self.nr_dqkv, 1, 3 * D, 0, hv, 1, D, 0, self._view(ue + NRMS_W[0], 3 * D * D, (3 * D, D), grad=True), D, 0,
                        None, 0, 3 * D, D, BU, ksplit=self.KS)
            self._c("tnr_colsum", self.nr_dqkv, 3 * D, T.F32, BU, 3 * D, self._view(ue + NRMS_B[0], 3 * D, (3 * D,), grad=True),
                    self.cs_part, 0)
            self._sgemm(self.nr_dqkv, 3 * D, 1, 0, wqkv, 1, D, 0, self.nr_dhv, D, 0, None, 0, BU, D, 3 * D)
            T.call("tnr_user_blend_bwd", self.nr_dhv, self.mask, hidx, ulm, dS, self.user_part[:, 2 * Qu:], self.user_part.shape[1],
                   B, U, D)
        ps = self.user_part.shape[1]
        rbh.add(self.user_part, B, ps, ps, self._view(ue + "attn.att_fc1.bias", ps, (ps,), grad=True))
        if self.plan is None:
            self.backward_encoder(dS[:N], N, after_bucket=after_bucket)
        else:
            p = self.plan
            T.call("tnr_segment_sum_rows", dS, p.order, p.seg, p.n_enc, D, self.dSv)
            self.backward_encoder(self.dSv, p.n_enc, after_bucket=after_bucket)

    def _transform_grads(self, Rt, rb, pend=None):
        """dW_i = dP_i^T X_i ; db_i = colsum(dP_i)   (transform_matrix, model_bert.py:278,283).  The bias column sums ride in the
        heads' batched reduction `rb` (flushed in backward_encoder, after the GEMM above has read dP: its first level sums in place)."""
        D, T_ = self.cfg.D, self.cfg.T
        dWt = self._view("transform_matrix.0.weight", T_ * D * D, (T_, D, D), grad=True)
        dbt = self._view("transform_matrix.0.bias", T_ * D, (T_, D), grad=True)
        dwt = (self.dP, 1, D, Rt * D, self.X, 1, D, self.X.stride(0), dWt, D, D * D, None, 0, D, D, Rt)
        if pend is None:
            self._sgemm(*dwt, batch=T_, ksplit=self.KS)
        else:          # the caller groups it with other independent GEMMs (one launch)
            pend.append(self._sgemm_problem(*dwt, batch=T_, ksplit=self.KS))
        for i in range(T_):
            rb.add(self.dP[i], Rt, D, D, dbt[i])

    def backward_encoder(self, dvec, N, acc=0, after_bucket=None, pend=None):
        """NewsEncoder backward for the N sequences of the last encode(): dvec (N,D) fp32 = d loss / d news vectors.
        acc=1 adds to the gradients already in flat_g (second pass over the same parameters, stage 1).
        pend: fp32 GEMM problems of the caller that depend on nothing computed here; they ride in the first grouped launch."""
        for _ in self.backward_encoder_steps(dvec, N, acc, after_bucket, pend=pend):
            pass

    def backward_encoder_steps(self, dvec, N, acc=0, after_bucket=None, defer=False, split_ffn=False, pend=None):
        """backward_encoder as a generator.  defer=True (stage 1): the weight gradients are not launched but collected in
        self._wg_defer, and the generator yields wherever they have to be on their way - after the pooling head, after every
        trainable layer (and, split_ffn, after its FFN block: the point a gradient bucket completes) - so that a caller running
        two passes over the same parameters in step can launch both passes' contributions to a weight as ONE chained problem."""
        cfg = self.cfg
        self._red_check()
        self._wg_defer = [] if defer else None
        L, D, H, I = cfg.L, cfg.D, cfg.H, cfg.I
        M = N * L
        g, gr = self.p, self.grads
        self._wg = None                    # nothing collected from an earlier, interrupted backward
        ds = self._dsite                   # sites of the forward call this backward belongs to (self.drop_cur)
        gi = self.ginv                     # parameter gradients below the pooling backward: 1 / loss scale on the way out
        # no bucket hook (one GPU): every partial sum of the backward in ONE batched reduction at its end
        one = after_bucket is None and self.merge_reductions
        rb = rb_heads = self.red.setdefault(("heads", acc, N, one), _ReduceBatch(self.dev))
        # dense + pooling of the news encoder
        wd = g(PFX + "dense.weight")
        self._sgemm_group(list(pend or []) + [
            self._sgemm_problem(dvec, 1, D, 0, self.nv, 1, H, 0, gr[PFX + "dense.weight"], H, 0, None, 0, D, H, N, ksplit=self.KS,
                                beta=float(acc)),
            self._sgemm_problem(dvec, D, 1, 0, wd, 1, H, 0, self.dnv, H, 0, None, 0, N, H, D, alpha=self.gscale)])   # loss scale enters here
        rb.add(dvec, N, D, D, gr[PFX + "dense.bias"], acc)       # column sums; in place, behind the two GEMMs that read dvec
        y = self.y_last
        if cfg.pooling == "att":
            self._attpool_bwd(y, self.e, self.dnv, self.dy2, self.dpre, self.dw2p, self.db2p, self.db1p, N)
            rb.add(self.dw2p, N, cfg.Qn, cfg.Qn, gr[PFX + "attn.att_fc2.weight"], acc, gi)
            rb.add(self.db2p, N, 1, 1, gr[PFX + "attn.att_fc2.bias"], acc, gi)
            rb.add(self.db1p, N, QPAD, QPAD, self._view(PFX + "attn.att_fc1.bias", QPAD, (QPAD,), grad=True), acc, gi)
        if not one or not cfg.trainable_layers:
            rb.flush()
        if cfg.pooling == "att":
            self._wgrad(self.dpre, y, self._view(PFX + "attn.att_fc1.weight", QPAD * H, (QPAD, H), grad=True), M, acc)
        if defer:
            yield "heads"
        if after_bucket:
            after_bucket(0)
        if not cfg.trainable_layers:
            self._wg_defer = None
            return
        if cfg.pooling == "att":
            self._gemm(self.dpre, self.sh_a1T, self.dy, M, res=self.dy2, flags=T.EPI_RES)
        else:
            self._c("tnr_pool_bwd", self.dnv, self.dy, N, L, H, int(cfg.pooling == "mean"))
        dy = self.dy
        bucket = 1
        for l in range(cfg.n_layers - 1, self.lo - 1, -1):
            names, sh, a = layer_param_order(l), self.sh[l], self.act[l - self.lo]
            tr = l in cfg.trainable_layers
            x_in = self.x_in[l]
            # bias gradients ride along: dx column sums from LayerNorm backward, the dgrad epilogue, attention backward;
            # all partial sums of the layer are reduced by one launch at the end (fixed order)
            # two gradient buckets per layer, in completion order: the FFN block (names[10:16]) is final after the W1
            # weight gradient, the attention block (names[0:10]) at the end of the layer -- the last all-reduce of a step
            # (attention block of layer lo) is then a third of a layer instead of a whole one
            # (without a bucket hook - one GPU - nothing waits for any bucket: the partial sums of all layers and of the heads go
            # into one batched reduction at the end of the backward)
            self._wg = [] if (tr and self.group_wgrad and not defer) else None
            rba = (rb_heads if one else self.red.setdefault((l, acc, N, "att"), _ReduceBatch(self.dev))) if tr else None
            rb = (rb_heads if one else self.red.setdefault((l, acc, N, "ffn"), _ReduceBatch(self.dev))) if tr else None
            P = self.lpart.get(l)
            nblk = T.query("tnr_ln_bwd_blocks", M)
            # with dropout behind the two output Linears the LayerNorm backward has two outputs: dx for the residual branch and
            # dx * mask / (1 - p) = the Linear's output gradient (its weight gradient, dgrad and -- through the partials -- bias)
            dF, dO, dPb = ds(T.DROP_FFN_OUT, l), ds(T.DROP_ATTN_OUT, l), ds(T.DROP_PROB, l)
            dypre_lin = self.dyprem if dF else self.dypre
            dh1pre_lin = self.dh1prem if dO else self.dh1pre
            lnargs = (dy, a["ypre"], a["st2"], g(names[14]), self.dypre, None, None, None, (P["ln_part"] if tr else None), M, H)
            self._c("tnr_ln_bwd_do", *lnargs, self.dyprem, dF) if dF else self._c("tnr_ln_bwd", *lnargs)
            if tr:
                rb.add(P["ln_part"], nblk, 3 * H, 2 * H, self._view(names[14], 2 * H, (2 * H,), grad=True), acc, gi)   # [dgamma | dbeta]
                rb.add(P["ln_part"][2 * H:], nblk, 3 * H, H, gr[names[13]], acc, gi)                                 # output.dense.bias
                self._wgrad(dypre_lin, a["g"], gr[names[12]], M, acc)
            fused_cs = tr and M > 128            # the column-sum epilogue needs more than one 128-row strip
            self._gemm(dypre_lin, sh["w2T"], self.du, M, aux=a["u"], flags=T.EPI_MULDGELU | (T.EPI_COLSUM if fused_cs else 0),
                       colsum=P["gcs_part"] if fused_cs else None)
            if tr:
                if fused_cs:
                    rb.add(P["gcs_part"], self._q("tnr_gemm_colsum_rows", M), I, I, gr[names[11]], acc, gi)
                else:
                    self._c("tnr_colsum", self.du, I, T.BF16, M, I, P["cs_tmp"][:I], self.cs_part, 0)
                    rb.add(P["cs_tmp"], 1, I, I, gr[names[11]], acc, gi)
                self._wgrad(self.du, a["h1"], gr[names[10]], M, acc)
                if not one:
                    rb.flush()
                if defer and split_ffn:
                    yield (l, "ffn")
                if after_bucket:
                    if self._wg is not None:
                        self._wgrad_flush()            # the FFN block's two gradients: its bucket goes out now
                    after_bucket(bucket)
                    bucket += 1
            self._gemm(self.du, sh["w1T"], self.dh1, M, res=self.dypre, flags=T.EPI_RES)
            lnargs = (self.dh1, a["h1pre"], a["st1"], g(names[8]), self.dh1pre, None, None, None, (P["ln_part1"] if tr else None), M, H)
            self._c("tnr_ln_bwd_do", *lnargs, self.dh1prem, dO) if dO else self._c("tnr_ln_bwd", *lnargs)
            if tr:
                rba.add(P["ln_part1"], nblk, 3 * H, 2 * H, self._view(names[8], 2 * H, (2 * H,), grad=True), acc, gi)
                rba.add(P["ln_part1"][2 * H:], nblk, 3 * H, H, gr[names[7]], acc, gi)                               # attention.output.dense.bias
                self._wgrad(dh1pre_lin, a["ctx"], gr[names[6]], M, acc)
            self._gemm(dh1pre_lin, sh["oT"], self.dctx, M)
            if self._attn_bwd(a["qkv"], a["ctx"], a["lse"], self.dctx, self.dqkv, (P["qkvb_part"] if tr else None), N, dPb):
                if tr:
                    rba.add(P["qkvb_part"], N, 3 * H, 3 * H, self._view(names[3], 3 * H, (3 * H,), grad=True), acc, gi)
            elif tr:
                self._c("tnr_colsum", self.dqkv, 3 * H, T.BF16, M, 3 * H, P["cs_tmp2"][:3 * H], self.cs_part, 0)
                rba.add(P["cs_tmp2"], 1, 3 * H, 3 * H, self._view(names[3], 3 * H, (3 * H,), grad=True), acc, gi)
            if tr:
                self._wgrad(self.dqkv, x_in, self._view(names[0], 3 * H * H, (3 * H, H), grad=True), M, acc)
                if self._wg is not None:               # all four of the layer (two under a bucket hook), before the next layer overwrites their operands
                    self._wgrad_flush()
                    self._wg = None
                if defer:
                    yield (l, "att")
                if not one:
                    rba.flush()
            if l > self.lo:
                nxt = self.dy2 if dy is self.dy else self.dy
                self._gemm(self.dqkv, sh["qkvT"], nxt, M, res=self.dh1pre, flags=T.EPI_RES)
                dy = nxt
            if tr and after_bucket:
                after_bucket(bucket)
                bucket += 1
        if one:
            rb_heads.flush()
        self._wg_defer = None

    def grad(self, name):
        """Gradient of a trainable parameter (a view into flat_g; the fp16 loss scale never reaches it)."""
        return self.grads[name]

    def bucket_ranges(self):
        """Contiguous [start, end) slices of flat_g in the order their gradients complete."""
        out = [(self.off(PFX + ("attn.att_fc1.weight" if self.cfg.pooling == "att" else "dense.weight")), self.n_train)]
        for l in sorted(self.cfg.trainable_layers, reverse=True):
            names = layer_param_order(l)
            a0, f0 = self.off(names[0]), self.off(names[10])
            e = self.off(names[-1]) + self.slot[names[-1]][2]
            out.append((f0, _rup(e, 64)))          # FFN block: intermediate.dense .. output.LayerNorm
            out.append((a0, f0))                   # attention block: q/k/v .. attention.output.LayerNorm
        return out

    # ------------------------------------------------------------------ optimiser
    def step(self, lr, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-8, lr_bert=None, amsgrad=True, lr_news_head=None, sync=None):
        """torch.optim.Adam(amsgrad=True).step() (run.py:134,195) + refresh of the 16-bit weight copies.
        lr_bert / lr_news_head: learning rates of the encoder layers / of the news encoder's pooling + dense when they
        differ (PLM-NR/run.py:104-106: {'params': pretrained, 'lr': pretrain_lr}, {'params': rest, 'lr': lr}; the notebooks
        use 1e-6 for bert_model and 1e-5 for the rest).  amsgrad=False: plain Adam (Post-train_KD.ipynb cell 18).
        sync: a dist.GradSync with all-reduces in flight -- the update then runs bucket by bucket in completion order, each
        slice behind its own collective only (elementwise optimiser: the same bits as one launch over everything)."""
        sc = self.scaler
        guard, stamp = None, 0
        if sc.enabled:
            used = sc.mult                   # what this step's backward ran with
            sc.poll(self)                    # overflows of steps two or more back: scale halved, Adam's count corrected
            sc.stamp += 1
            guard, stamp = sc.guard, sc.stamp
        self.step_count += 1
        head0 = self.off(PFX + ("attn.att_fc1.weight" if self.cfg.pooling == "att" else "dense.weight"))   # end of the BERT layers
        e = self.off(PFX + "dense.bias") + self.slot[PFX + "dense.bias"][2]
        rest0 = min(_rup(e, 64), self.n_train)   # user encoders / transform matrices start here
        lb = lr if lr_bert is None else lr_bert
        lh = lr if lr_news_head is None else lr_news_head
        ranges = []
        for lo_, hi_, rate in ((0, head0, lb), (head0, rest0, lh), (rest0, self.n_train, lr)):     # neighbours with one rate: one launch
            if ranges and ranges[-1][2] == rate and ranges[-1][1] == lo_:
                ranges[-1] = (ranges[-1][0], hi_, rate)
            elif hi_ > lo_:
                ranges.append((lo_, hi_, rate))
        def launch(lo_, hi_, rate):
            if hi_ > lo_:
                args = (self.flat[True][lo_:hi_], self.flat_g[lo_:hi_], self.adam_m[lo_:hi_], self.adam_v[lo_:hi_],
                        self.adam_vmax[lo_:hi_] if amsgrad else None, hi_ - lo_, self.step_count, rate, beta1, beta2, eps, grad_scale)
                if guard is None:
                    T.call("tnr_amsgrad_step", *args)
                else:
                    T.call("tnr_amsgrad_step_guarded", *args, guard, stamp, sc.skipped)

        if guard is not None:
            # the WHOLE (reduced) gradient decides before any slice is updated - identically on every rank, since inf / nan
            # survive the all-reduce.  Under data parallelism each bucket is scanned as soon as ITS all-reduce has landed (the
            # scans hide under the collectives still in flight; what lies between buckets are alignment gaps, zero gradients); only
            # the last bucket's scan, the one-thread commit and the update itself (79 us at the headline) stay behind the last
            # collective.  (bf16 has no guard and updates bucket by bucket, below.)
            if sync is not None and sync.pending:
                for b, (s_, e_) in enumerate(sync.ranges):
                    sync.wait_bucket(b)
                    T.call("tnr_grad_nonfinite_scan", self.flat_g[s_:e_], e_ - s_, guard, stamp)
                T.call("tnr_grad_nonfinite_commit", guard, stamp)
            else:
                if sync is not None:
                    sync.wait()
                T.call("tnr_grad_nonfinite", self.flat_g, self.n_train, guard, stamp)
            for lo_, hi_, rate in ranges:
                launch(lo_, hi_, rate)
            sc.record(used)
        elif sync is not None and sync.pending:
            done = []
            for b, (s_, e_) in enumerate(sync.ranges):       # completion order of backward = launch order of the all-reduces
                sync.wait_bucket(b)
                for lo_, hi_, rate in ranges:
                    launch(max(lo_, s_), min(hi_, e_), rate)
                done.append((s_, e_))
            pos = 0                                          # anything outside the buckets (alignment gaps: zero gradients)
            for s_, e_ in sorted(done) + [(self.n_train, self.n_train)]:
                for lo_, hi_, rate in ranges:
                    launch(max(lo_, pos), min(hi_, s_), rate)
                pos = max(pos, e_)
        else:
            if sync is not None:
                sync.wait()
            for lo_, hi_, rate in ranges:
                launch(lo_, hi_, rate)
        self.refresh_shadows(all_layers=False)
