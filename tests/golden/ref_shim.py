"""Harness-only: import the reference (read-only, /root/reference) in THIS container.

Never runs on the GPU box and nothing here is shipped as product code: it only
exists so that tests/golden/make_golden.py can execute the reference's own
Python and record input/output vectors.  The shims adapt
the reference's pinned third-party versions (transformers 3.0.2, TF 1.15) to
what this image has; they do not touch the reference's arithmetic.
"""
import importlib
import json
import os
import sys
import tempfile
import types

REF_ROOT = "/root/reference"


def _install_shims():
    mb = importlib.import_module("transformers.models.bert.modeling_bert")
    sys.modules["transformers.modeling_bert"] = mb
    tb = importlib.import_module("transformers.models.bert.tokenization_bert")
    if not hasattr(tb, "whitespace_tokenize"):
        tb.whitespace_tokenize = lambda t: t.split()
    sys.modules["transformers.tokenization_bert"] = tb
    import transformers.modeling_utils as mu
    for n, v in (("cached_path", None), ("TF2_WEIGHTS_NAME", "x"), ("TF_WEIGHTS_NAME", "y")):
        if not hasattr(mu, n):
            setattr(mu, n, v)
    tf = types.ModuleType("tensorflow")
    tf.io = types.SimpleNamespace(gfile=types.SimpleNamespace(
        GFile=open, exists=os.path.exists, listdir=os.listdir, isdir=os.path.isdir))
    sys.modules["tensorflow"] = tf


def load_reference(tree="Tiny-NewsRec"):
    """Returns a namespace of the reference's modules (model_bert, dataloader, ...)."""
    _install_shims()
    path = os.path.join(REF_ROOT, tree)
    for m in ("utils", "model_bert", "model_bert_2", "preprocess", "dataloader", "streaming",
              "parameters", "tnlrv3", "tnlrv3.modeling"):
        sys.modules.pop(m, None)
    # the build's own tree mirrors the reference's module names (incl. a regular package `tnlrv3`, which would win
    # over the reference's namespace package wherever it sits on sys.path): take it off the path while importing
    hidden = [p for p in sys.path if os.path.basename(os.path.normpath(p)) == "tiny-newsrec_amd"]
    saved_path = list(sys.path)
    sys.path[:] = [p for p in sys.path if p not in hidden]
    for m in [m for m in sys.modules if m == "tnlrv3" or m.startswith("tnlrv3.")]:
        sys.modules.pop(m)
    sys.path.insert(0, path)
    try:
        import utils, model_bert, preprocess, dataloader, streaming  # noqa: E401
        from tnlrv3 import modeling as M
        from tnlrv3.tokenization_tnlrv3 import TuringNLRv3Tokenizer as T
        from tnlrv3 import convert_state_dict as CSD
    finally:
        sys.path[:] = saved_path
    M.TuringNLRv3PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    # no unilm2 .bin offline; .eval() reproduces HF's post-load mode
    M.TuringNLRv3ForSequenceClassification.from_pretrained = classmethod(
        lambda cls, path, config=None, **kw: cls(config).eval())
    if not getattr(T, "_tnr_patched", False):
        _c = T.__call__
        T.__call__ = lambda self, text, max_length=None, pad_to_max_length=False, truncation=False, **kw: _c(
            self, text, max_length=max_length,
            padding="max_length" if pad_to_max_length else False, truncation=truncation, **kw)
        T._tnr_patched = True
    return types.SimpleNamespace(utils=utils, model_bert=model_bert, preprocess=preprocess,
                                 dataloader=dataloader, streaming=streaming, modeling=M, convert_state_dict=CSD,
                                 tokenizer_cls=T, path=path)


def write_config(cfg, dirname=None):
    """Write a tnlrv3 config json (any sizes) and return its path."""
    d = dirname or tempfile.mkdtemp(prefix="tnr_cfg_")
    p = os.path.join(d, "config.json")
    with open(p, "w") as f:
        json.dump(cfg, f)
    return p


BASE_CFG = dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
                hidden_size=768, initializer_range=0.02, intermediate_size=3072,
                max_position_embeddings=512, num_attention_heads=12, num_hidden_layers=12,
                type_vocab_size=2, vocab_size=30522, rel_pos_bins=32, max_rel_pos=128)


def make_args(**over):
    """Plain attribute bag with the flags the reference reads (values from demo.sh)."""
    a = dict(model_type="tnlrv3", pooling="att", model="NAML", news_dim=256,
             news_query_vector_dim=200, user_query_vector_dim=200, num_attention_heads=16,
             user_log_length=50, user_log_mask=False, npratio=4, batch_size=4,
             num_teacher_layers=12, num_student_layers=4, num_teachers=4, temperature=1.0,
             coef=0.2, num_words_title=30, shuffle_buffer_size=10000,
             model_name="unused.bin",
             config_name=os.path.join(REF_ROOT, "Tiny-NewsRec/tnlrv3/config/tnlrv3-base-uncased-config.json"),
             tokenizer_name=os.path.join(REF_ROOT, "Tiny-NewsRec/tnlrv3/tokenizer/tnlrv3-base-uncased-vocab.txt"))
    a.update(over)
    return types.SimpleNamespace(**a)
