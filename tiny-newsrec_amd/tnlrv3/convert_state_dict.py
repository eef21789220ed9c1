"""Import of pretrained UniLMv2 weights.

* load_model(state_dict)              - tnlrv3/convert_state_dict.py:39-71: the published unilm2 checkpoint stores one
  fused `qkv_linear.weight` (3H, H) and `q_bias` / `v_bias` per layer and keeps the rel-pos bias under `bert.encoder.`;
  the model wants separate query/key/value Linear parameters (key bias = zeros) and `bert.rel_pos_bias.weight`.
* resize_position_embeddings(...)     - tnlrv3/modeling.py:90-118: grow (optionally tiling the old table over the new
  range) or truncate `bert.embeddings.position_embeddings.weight` to config.max_position_embeddings.
* student_state_from_pretrained(...)  - what `model_class.from_pretrained(args.model_name, config=...)` leaves in
  `student.news_encoder.bert_model` (model_bert.py:109-114): converted keys under the module prefix, layers beyond
  num_hidden_layers and heads the model does not have (cls.*, seq2seq) dropped as HF's non-strict load does.

Host-side dictionary work on CPU tensors (runs once at start-up, not on the per-step path)."""
import os
import re

import torch

# suffix in the checkpoint -> [(suffix in the model, transform)]
_FLAT = lambda v: v.reshape(-1)
_SELF = "attention.self."


def _split3(v):
    rows = v.shape[0]
    if rows % 3:
        raise ValueError("qkv_linear.weight with %d rows is not a stack of three square blocks" % rows)
    return [v[i * (rows // 3):(i + 1) * (rows // 3)] for i in range(3)]


def load_model(state_dict):
    """unilm2 checkpoint dict -> dict with the model's parameter names.  Values are views / reshapes of the inputs
    (no copies) except the all-zero key bias."""
    out = {}
    for key, value in state_dict.items():
        if key.endswith(_SELF + "qkv_linear.weight"):
            stem = key[:-len("qkv_linear.weight")]
            for name, block in zip(("query", "key", "value"), _split3(value)):
                out[stem + name + ".weight"] = block
        elif key.endswith(_SELF + "q_bias"):
            out[key[:-len("q_bias")] + "query.bias"] = _FLAT(value)
        elif key.endswith(_SELF + "v_bias"):
            stem = key[:-len("v_bias")]
            out[stem + "value.bias"] = _FLAT(value)
            out[stem + "key.bias"] = torch.zeros_like(_FLAT(value))       # softmax removes a key bias (SURVEY appendix v)
        elif key == "bert.encoder.rel_pos_bias.weight":
            out["bert.rel_pos_bias.weight"] = value
        else:
            out[key] = value
    return out


state_dict_convert = {"tnlrv3": load_model}

POS_KEY = "bert.embeddings.position_embeddings.weight"


def resize_position_embeddings(table, max_position_embeddings, initializer_range=0.02, reuse_position_embedding=None,
                               generator=None):
    """-> (max_position_embeddings, H) fp32.  Growing: rows [0, old) are the old table; with
    reuse_position_embedding the old table is tiled over the whole new range, otherwise the new rows are
    N(0, initializer_range) (the reference draws them from torch's global RNG; pass `generator` to pin them).
    Shrinking: the first rows.  Same size: returned unchanged."""
    old, H = table.shape
    new = int(max_position_embeddings)
    if new == old:
        return table
    if new < old:
        return table[:new].to(torch.float32).clone()
    out = torch.empty((new, H), dtype=torch.float32).normal_(0.0, initializer_range, generator=generator)
    span = new if reuse_position_embedding else old
    for start in range(0, span, old):
        n = min(old, span - start)
        out[start:start + n] = table[:n]
    return out


_LAYER = re.compile(r"^bert\.encoder\.layer\.(\d+)\.")


def student_state_from_pretrained(state_dict, wanted, n_layers, max_position_embeddings, initializer_range=0.02,
                                  reuse_position_embedding=None, prefix="student.news_encoder.bert_model.",
                                  generator=None):
    """Raw unilm2 checkpoint -> {prefix + key: tensor} for the keys in `wanted` (the model's schema) that the
    checkpoint provides; returns (state, missing, unexpected) like a non-strict load_state_dict."""
    sd = load_model(state_dict)
    if POS_KEY in sd:
        sd[POS_KEY] = resize_position_embeddings(sd[POS_KEY], max_position_embeddings, initializer_range,
                                                 reuse_position_embedding, generator)
    state, unexpected = {}, []
    for key, value in sd.items():
        m = _LAYER.match(key)
        full = prefix + key
        if (m and int(m.group(1)) >= n_layers) or full not in wanted:
            unexpected.append(key)
            continue
        if tuple(value.shape) != tuple(wanted[full]):
            raise ValueError("%s: checkpoint shape %s, model %s" % (key, tuple(value.shape), tuple(wanted[full])))
        state[full] = value.to(torch.float32)
    missing = [k for k in wanted if k.startswith(prefix) and k not in state]
    return state, missing, unexpected


def read_checkpoint(path):
    """A file, or a directory holding pytorch_model.bin (tnlrv3/modeling.py:69-77)."""
    if os.path.isdir(path):
        path = os.path.join(path, "pytorch_model.bin")
    if not os.path.isfile(path):
        raise FileNotFoundError("pretrained checkpoint not found: %s" % path)
    return torch.load(path, map_location="cpu")
