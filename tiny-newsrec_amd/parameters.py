"""Flag surface of the reference's run.py (Tiny-NewsRec/parameters.py:8-115): every flag keeps its name,
type and default so demo.sh command lines work unchanged.  Flags the reference parses but never reads
 are accepted and ignored here too.  Added flags default to the reference behaviour."""
import argparse
import logging

import utils

_B = utils.str2bool
# (name, type, default[, extra kwargs])
_FLAGS = [
    ("mode", str, "train", dict(choices=["train", "test", "get_teacher_emb"])),
    ("train_data_dir", str, "../MIND/MINDlarge_train"),
    ("test_data_dir", str, "../MIND/MINDlarge_test"),
    ("filename_pat", str, "behaviors_np4_*.tsv"),
    ("model_dir", str, "./model"),
    ("batch_size", int, 32),
    ("npratio", int, 4),
    ("enable_gpu", _B, True),
    ("enable_hvd", _B, True),            # kept: True = data-parallel over RCCL (horovod is gone)
    ("enable_shuffle", _B, True),
    ("shuffle_buffer_size", int, 10000),
    ("num_workers", int, 4),
    ("filter_num", int, 3),
    ("log_steps", int, 100),
    ("epochs", int, 1),
    ("lr", float, 0.0001),
    ("num_words_title", int, 20),
    ("num_words_abstract", int, 50),
    ("num_words_body", int, 100),
    ("user_log_length", int, 50),
    ("word_embedding_dim", int, 300),
    ("glove_embedding_path", str, "./glove.840B.300d.txt"),
    ("freeze_embedding", _B, False),
    ("news_dim", int, 64),
    ("news_query_vector_dim", int, 200),
    ("user_query_vector_dim", int, 200),
    ("num_attention_heads", int, 20),
    ("user_log_mask", _B, True),
    ("drop_rate", float, 0.2),
    ("save_steps", int, 1000),
    ("max_steps_per_epoch", int, 1000000),
    ("load_ckpt_name", str, None, dict(help="choose which ckpt to load and test")),
    ("apply_bert", _B, False),
    ("model_type", str, "bert"),
    ("do_lower_case", _B, True),
    ("model_name", str, "../bert-base-uncased/pytorch_model.bin"),
    ("config_name", str, "../bert-base-uncased/config.json"),
    ("tokenizer_name", str, "../bert-base-uncased/vocab.txt"),
    ("num_hidden_layers", int, 8),
    ("bert_trainable_layer", int, [], dict(nargs="+", choices=list(range(12)))),
    ("model", str, None),
    ("pooling", str, "att"),
    ("start_epoch", int, 0),
    ("use_pretrain_model", _B, False),
    ("pretrain_model_path", str, None),
    ("pretrain_lr", float, 0.00001),
    ("num_teacher_layers", int, 12),
    ("num_student_layers", int, 4),
    ("temperature", float, 1.0),
    ("coef", float, 1.0),
    ("tensorboard", str, None),
    ("teacher_ckpts", str, [], dict(nargs="+")),
    ("teacher_emb_paths", str, [], dict(nargs="+")),
    ("num_teachers", int, 4),
    # --- additions (defaults keep the reference behaviour)
    ("resident_tables", _B, True, dict(help="keep news_combined / teacher tables in HBM and ship indices only")),
    ("dtype", str, "bf16", dict(choices=["bf16", "fp16"], help="16-bit activation type of the HIP kernels")),
    ("synthetic", _B, False, dict(help="random-init weights + synthetic MIND-shaped data (no files needed)")),
]


def build_parser():
    ap = argparse.ArgumentParser()
    for f in _FLAGS:
        kw = dict(f[3]) if len(f) > 3 else {}
        ap.add_argument("--" + f[0], type=f[1], default=f[2], **kw)
    return ap


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    logging.info(args)
    return args


if __name__ == "__main__":
    print(parse_args())
