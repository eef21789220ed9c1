"""Module surface of Tiny-NewsRec/model_bert.py on the HIP engine: same class names, constructor and forward
signatures, return tuples and state_dict key schema, so run.py / checkpoints interchange.

    Model(args).forward(history, history_mask, candidate, label, teacher_history_embs, teacher_candidate_embs)
        -> (total_loss, distill_loss, emb_loss, target_loss, student_score)        model_bert.py:262-305

Every nn.Parameter is a view into the engine's flat fp32 buffers; total_loss.backward() runs the hand-written
HIP backward and leaves gradients in the parameters' .grad (views of the flat gradient buffer).  The module is
a thin shell: no arithmetic happens in torch, and construction fails if libtnr_hip.so is missing."""
import json
import logging

import torch
from torch import nn

import engine as E


def read_model_config(path, synthetic=False):
    """The tnlrv3 / BERT config json.  path None = UniLMv2-base sizes (tests, synthetic runs); a path that cannot be read
    raises, as config_class.from_pretrained does (model_bert.py:110) -- a mistyped --config_name must not silently train a
    768/12/3072 encoder.  Only --synthetic runs, which need no files at all, fall back (with a warning)."""
    if path is None:
        return {}
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError) as e:
        if synthetic:
            logging.warning("--config_name %s unreadable (%s): synthetic run, UniLMv2-base sizes assumed", path, e)
            return {}
        raise FileNotFoundError("--config_name %s cannot be read: %s" % (path, e)) from e


def engine_config_from_args(args, num_teachers=None, is_teacher=False):
    """parameters.py flags -> EngineConfig.  Model sizes come from the tnlrv3 config json like
    config_class.from_pretrained(args.config_name, num_hidden_layers=...) at model_bert.py:110-113."""
    cfg = read_model_config(getattr(args, "config_name", None), getattr(args, "synthetic", False))
    nl = args.num_teacher_layers if is_teacher else args.num_student_layers
    T_ = args.num_teachers if num_teachers is None else num_teachers
    return E.EngineConfig(
        n_layers=nl, trainable_layers=tuple(l for l in args.bert_trainable_layer if l < nl),
        hidden=cfg.get("hidden_size", 768), heads=cfg.get("num_attention_heads", 12), inter=cfg.get("intermediate_size", 3072),
        news_dim=args.news_dim, news_query=args.news_query_vector_dim, user_query=args.user_query_vector_dim,
        num_teachers=T_, user_log_length=args.user_log_length, npratio=args.npratio, num_words=args.num_words_title,
        user_log_mask=args.user_log_mask, temperature=args.temperature, coef=args.coef, vocab=cfg.get("vocab_size", 30522),
        max_pos=cfg.get("max_position_embeddings", 512), type_vocab=cfg.get("type_vocab_size", 2),
        ln_eps=cfg.get("layer_norm_eps", 1e-12), pooling=getattr(args, "pooling", "att"),
        pos_pad_id=cfg.get("pad_token_id", 1) if getattr(args, "tnr_model_type", "tnlrv3") == "roberta" else None,
        nrms_heads=getattr(args, "num_attention_heads", 0) if getattr(args, "model", "NAML") == "NRMS" else 0)


@torch.no_grad()
def reference_init(engine, seed=0):
    """The reference's construction-time distributions (not its RNG stream) for every parameter of `engine`: bert_model
    by TuringNLRv3PreTrainedModel._init_weights (tnlrv3/modeling.py:42-52: Linear / Embedding weights N(0, 0.02),
    LayerNorm (1, 0), Linear biases 0); heads by nn.Linear's default U(+-1/sqrt(fan_in)) for weight and bias
    (model_bert.py:12-13, 115; UserEncoder :150); pad_doc U(-1, 1) (:151-153); transform_matrix and the NRMS
    projections Xavier-uniform (:81, :258-260; transform bias 0)."""
    gen = torch.Generator().manual_seed(int(seed))
    for k, p in engine.params.items():
        t = torch.empty(p.shape, dtype=torch.float32)
        if ".bert_model." in k:
            if "LayerNorm.weight" in k:
                t.fill_(1.0)
            elif k.endswith("bias"):
                t.zero_()
            else:
                t.normal_(0.0, 0.02, generator=gen)
        elif k.endswith("pad_doc"):
            t.uniform_(-1.0, 1.0, generator=gen)
        elif k.startswith("transform_matrix.") or ("multi_head_self_attn" in k and k.endswith("weight")):
            if k.endswith("bias"):
                t.zero_()
            else:
                b = (6.0 / (p.shape[0] + p.shape[1])) ** 0.5
                t.uniform_(-b, b, generator=gen)
        else:                                              # nn.Linear default; the bias uses its weight's fan_in
            fan_in = p.shape[1] if p.dim() == 2 else engine.params[k[:-len("bias")] + "weight"].shape[1]
            b = 1.0 / fan_in ** 0.5
            t.uniform_(-b, b, generator=gen)
        p.copy_(t)
    engine.refresh_shadows(all_layers=True)


def load_pretrained_into(engine, path, seed=0, allow_missing=False):
    """model_class.from_pretrained(args.model_name, config=...) (model_bert.py:114) for the student's encoder of
    `engine`: convert the unilm2 checkpoint (tnlrv3/convert_state_dict.py), fit the position table, keep the first
    n_layers layers.  -> (missing, unexpected).  A path that does not exist raises like from_pretrained does, unless
    allow_missing (--synthetic / --allow_random_init): then None is returned and the initialisation stays."""
    import os
    if not path:
        return None
    if not os.path.exists(path):
        if allow_missing:
            logging.warning("pretrained encoder %s not found: keeping the construction-time initialisation", path)
            return None
        raise FileNotFoundError("--model_name %s does not exist (pass --allow_random_init True to train from the "
                                "construction-time initialisation)" % path)
    from tnlrv3 import convert_state_dict as C
    cfg = engine.cfg
    wanted = {k: tuple(v.shape) for k, v in engine.params.items()}
    state, missing, unexpected = C.student_state_from_pretrained(
        C.read_checkpoint(path), wanted, cfg.n_layers, cfg.max_pos, generator=torch.Generator().manual_seed(int(seed)))
    with torch.no_grad():
        for k, v in state.items():
            engine.params[k].copy_(v)
    engine.refresh_shadows(all_layers=True)
    return missing, unexpected


def load_hf_bert_into(engine, path, allow_missing=False):
    """BertModel.from_pretrained / RobertaModel.from_pretrained (PLM-NR/model_bert.py:114, model_type bert / roberta): a
    transformers checkpoint (keys optionally prefixed "bert." / "roberta.", old LayerNorm.gamma / beta names) onto the
    engine's encoder; only the first n_layers layers are taken, like num_hidden_layers in the config does.  -> (missing,
    unexpected)."""
    import os
    if not path:
        return None
    if not os.path.exists(path):
        if allow_missing:
            logging.warning("pretrained encoder %s not found: keeping the construction-time initialisation", path)
            return None
        raise FileNotFoundError("--model_name %s does not exist (pass --allow_random_init True to train from the "
                                "construction-time initialisation)" % path)
    from tnlrv3 import convert_state_dict as C
    sd = C.read_checkpoint(path)
    sd = sd.get("model_state_dict", sd)
    wanted = {k[len(E.BERT):]: k for k in engine.params if k.startswith(E.BERT) and not k.endswith("rel_pos_bias.weight")}
    used, unexpected = set(), []
    with torch.no_grad():
        for k, v in sd.items():
            name = k
            for pre in ("bert.", "roberta."):
                if name.startswith(pre):
                    name = name[len(pre):]
            name = name.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
            tgt = wanted.get(name)
            if tgt is None or tuple(engine.params[tgt].shape) != tuple(v.shape):
                unexpected.append(k)
                continue
            engine.params[tgt].copy_(v)
            used.add(name)
    engine.refresh_shadows(all_layers=True)
    return sorted(set(wanted) - used), unexpected


class _Backward(torch.autograd.Function):
    """Bridges total_loss.backward() (run.py:194) to Engine.backward()."""

    @staticmethod
    def forward(ctx, anchor, model, total):
        ctx.model = model
        return total.clone()

    @staticmethod
    def backward(ctx, gout):
        m = ctx.model
        m.engine.backward(after_bucket=m._after_bucket)
        if m.strict_grad_scale:
            m.engine.flat_g.mul_(gout)          # honour a non-unit upstream gradient (costs one pass)
        m._bind_grads()
        return torch.zeros_like(m._anchor), None, None


class _Shell(nn.Module):
    """nn.Module tree whose leaves are parameters aliased to engine storage; built from dotted key names."""

    def _adopt(self, engine, prefix, rename=None):
        """rename(key without prefix) -> module path, or None to keep the parameter out of the module tree."""
        for key, tensor in engine.params.items():
            if not key.startswith(prefix):
                continue
            name = key[len(prefix):] if rename is None else rename(key[len(prefix):])
            if name is None:
                continue
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Shell())
                mod = mod._modules[p]
            prm = nn.Parameter(tensor, requires_grad=E.is_trainable(engine.cfg, key))
            mod.register_parameter(parts[-1], prm)


class Model(_Shell):
    def __init__(self, args, device=None, max_batch=None):
        super().__init__()
        self.args = args
        dev = device or ("cuda:%d" % torch.cuda.current_device())
        self.engine = E.Engine(engine_config_from_args(args), dev, max_batch=max_batch or args.batch_size,
                               dtype=getattr(args, "dtype", "fp16"))
        if type(self) is Model and getattr(args, "model_type", "tnlrv3") != "tnlrv3":
            # parameters.py defaults --model_type to "bert", demo.sh passes tnlrv3; the reference's Tiny-NewsRec tree cannot run
            # anything else (run.py:113-116 dereferences bert_model.bert, which BertModel / RobertaModel do not have)
            logging.warning("--model_type %s: Tiny-NewsRec's Model is the tnlrv3 (UniLM) encoder; bert / roberta exist on the "
                            "PLM-NR path only (--num_teachers 0)", args.model_type)
        self._adopt(self.engine, "")
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._after_bucket = None
        self.strict_grad_scale = False
        self._name_of = {id(p): k for k, p in self.named_parameters()}
        self.reset_parameters(getattr(args, "seed", 0) if getattr(args, "seed", None) is not None else 0)
        self.pretrained_report = self.load_pretrained(getattr(args, "model_name", None))

    def reset_parameters(self, seed=0):
        reference_init(self.engine, seed)

    def load_pretrained(self, path):
        """-> (missing, unexpected); None only for path None or (--synthetic / --allow_random_init) a missing file."""
        a = self.args
        allow = bool(getattr(a, "synthetic", False) or getattr(a, "allow_random_init", False))
        rep = load_pretrained_into(self.engine, path, allow_missing=allow)
        if rep is not None:      # the reference prints what it loaded / initialised (tnlrv3/modeling.py:53-130)
            logging.info("pretrained encoder %s: %d missing keys %s, %d unexpected keys %s", path, len(rep[0]), list(rep[0])[:8],
                         len(rep[1]), list(rep[1])[:8])
        return rep

    def _engine_key(self, name):
        return name

    # reference-compatible views --------------------------------------------------------------------
    def named_trainable(self):
        return [(k, p) for k, p in self.named_parameters() if p.requires_grad]

    def _bind_grads(self):
        for k, p in self.named_parameters():
            if p.requires_grad and p.grad is None:
                p.grad = self.engine.grads[self._engine_key(k)]      # true gradients in both 16-bit modes

    def load_state_dict(self, sd, strict=True):
        out = super().load_state_dict(sd, strict=strict)
        self.engine.refresh_shadows(all_layers=True)       # bf16 copies + rel-pos table follow the fp32 weights
        return out

    def forward(self, history, history_mask, candidate, label, teacher_history_embs, teacher_candidate_embs):
        eng = self.engine
        if isinstance(history, tuple) and len(history) == 4:      # resident-mode IndexBatch
            raise TypeError("use forward_indexed for resident-mode batches")
        losses, score = eng.forward(history, history_mask, candidate, label, list(teacher_history_embs),
                                    list(teacher_candidate_embs))
        return self._pack(losses, score)

    def forward_indexed(self, news_combined, hist_idx, history_mask, cand_idx, label, teacher_tables, plan=None):
        losses, score = self.engine.forward_indexed(news_combined, hist_idx, history_mask, cand_idx, label, teacher_tables,
                                                    plan)
        return self._pack(losses, score)

    def _pack(self, losses, score):
        total = _Backward.apply(self._anchor, self, self.engine.total_loss())
        return total, losses[0], losses[2], losses[1], score


class ModelBert(Model):
    """PLM-NR/model_bert.py:178-207 (BASELINE configs[0]/[1]: the teacher / baseline fine-tuning tree): the same news and
    user encoders with a plain cross-entropy objective,
        ModelBert(args).forward(history, history_mask, candidate, label) -> (loss, score)
    and state_dict keys without the "student." prefix (news_encoder.*, user_encoder.*), so PLM-NR checkpoints load
    directly and what it saves is what Tiny-NewsRec's get_teacher_emb / teacher_ckpts read (run.py:61-70, 382-460).
    args.num_hidden_layers sizes the encoder (PLM-NR/model_bert.py:109-111).
    args.model_type (PLM-NR/utils.py:17-21, model_bert.py:109): 'tnlrv3' (demo.sh), or 'bert' / 'roberta' = transformers
    BertModel / RobertaModel as the encoder: the same layers without the rel-pos bias (its table is held at zero and, like the
    sequence-classification head, is not part of the module tree), parameters directly under bert_model.* as BertModel names
    them (PLM-NR/run.py:119-124 unfreezes bert_model.encoder.layer[i]), RoBERTa with its own position rule."""

    def __init__(self, args, device=None, max_batch=None):
        import types
        a = types.SimpleNamespace(**vars(args))
        a.num_student_layers = getattr(args, "num_hidden_layers", getattr(args, "num_student_layers", 12))
        a.num_teachers, a.temperature, a.coef = 0, 1.0, 1.0
        mt = getattr(args, "model_type", "tnlrv3")
        if mt not in ("tnlrv3", "bert", "roberta"):
            raise KeyError("--model_type %s: MODEL_CLASSES has tnlrv3, bert, roberta (PLM-NR/utils.py:17-21)" % mt)
        self.model_type = a.tnr_model_type = mt
        super().__init__(a, device, max_batch)

    def _rename(self, name):
        if self.model_type == "tnlrv3":
            return name
        if name.endswith("rel_pos_bias.weight") or ".bert_model.classifier." in name:
            return None
        return name.replace("news_encoder.bert_model.bert.", "news_encoder.bert_model.")

    def _adopt(self, engine, prefix):
        super()._adopt(engine, "student.", self._rename)      # module tree / parameter names drop the prefix

    def _engine_key(self, name):
        if self.model_type != "tnlrv3":
            name = name.replace("news_encoder.bert_model.", "news_encoder.bert_model.bert.")
        return "student." + name

    def reset_parameters(self, seed=0):
        super().reset_parameters(seed)
        if self.model_type != "tnlrv3":                # no relative-position bias in BertModel / RobertaModel
            self.engine.params[E.BERT + "rel_pos_bias.weight"].zero_()
            self.engine.refresh_rel()

    def load_state_dict(self, sd, strict=True):
        if self.model_type != "tnlrv3":                # transformers 3.0.2 keeps position_ids as a persistent buffer
            sd = {k: v for k, v in sd.items() if not k.endswith("embeddings.position_ids")}
        return super().load_state_dict(sd, strict=strict)

    def load_pretrained(self, path):
        if self.model_type == "tnlrv3":
            return super().load_pretrained(path)
        a = self.args
        allow = bool(getattr(a, "synthetic", False) or getattr(a, "allow_random_init", False))
        rep = load_hf_bert_into(self.engine, path, allow_missing=allow)
        if rep is not None:
            logging.info("pretrained %s encoder %s: %d missing keys %s, %d unexpected keys %s", self.model_type, path, len(rep[0]),
                         list(rep[0])[:8], len(rep[1]), list(rep[1])[:8])
        return rep

    def forward(self, history, history_mask, candidate, label):
        losses, score = self.engine.forward(history, history_mask, candidate, label)
        total = _Backward.apply(self._anchor, self, self.engine.total_loss())
        return total, score

    def forward_indexed(self, news_combined, hist_idx, history_mask, cand_idx, label, plan=None):
        losses, score = self.engine.forward_indexed(news_combined, hist_idx, history_mask, cand_idx, label, None, plan)
        total = _Backward.apply(self._anchor, self, self.engine.total_loss())
        return total, score


class TnrAdam:
    """optim.Adam(model.parameters(), lr, amsgrad=True) of run.py:134 on the engine's flat buffers
    (one fused kernel + bf16-copy refresh); zero_grad / step keep the reference's call order (run.py:193-195)."""

    def __init__(self, model, lr, grad_sync=None, pretrain_lr=None, pretrained_heads=False):
        """pretrain_lr: PLM-NR/run.py:104-106 - the "pretrained" parameters step with their own rate: the encoder layers,
        and (pretrained_heads) the news encoder's pooling + dense when they came from the first-stage checkpoint too."""
        self.model, self.lr, self.grad_sync, self.pretrain_lr = model, lr, grad_sync, pretrain_lr
        self.pretrained_heads = pretrained_heads

    def zero_grad(self):
        pass                      # every backward overwrites the whole flat gradient buffer

    def step(self):
        scale = self.grad_sync.scale if self.grad_sync is not None else 1.0
        # the engine waits for the gradient all-reduces bucket by bucket and updates each slice behind its own collective
        self.model.engine.step(self.lr, grad_scale=scale, lr_bert=self.pretrain_lr,
                               lr_news_head=self.pretrain_lr if self.pretrained_heads else None, sync=self.grad_sync)
