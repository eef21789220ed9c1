"""news.tsv -> news_index and the padded token tables (layout of Tiny-NewsRec/preprocess.py:15-66), TF-free.

Tokenisation itself is one-off host work and out of scope for kernels: any callable
`tokenizer(text, max_length=, padding="max_length", truncation=True) -> {"input_ids", "attention_mask"}` works,
by default transformers' BertTokenizer on the reference's vocab file.  What matters downstream is the layout:
news ids are numbered 1..n in file order, row 0 of every table is the all-zero pad news, tables are int32."""
import numpy as np


def make_tokenizer(args):
    from transformers import BertTokenizer
    return BertTokenizer(vocab_file=args.tokenizer_name, do_lower_case=True)


def _number(table, key):
    return table.setdefault(key, len(table) + 1)


def read_news_bert(news_path, args, mode="train", tokenizer=None):
    assert mode in ("train", "test"), "Wrong mode!"
    encode = tokenizer or make_tokenizer(args)
    width = args.num_words_title
    news, news_index, categories, subcategories = {}, {}, {}, {}
    with open(news_path, "r", encoding="utf-8") as fh:
        for row in fh:
            doc_id, cat, subcat, title = row.rstrip("\n").split("\t")[:4]
            _number(news_index, doc_id)
            tokens = encode(title.lower(), max_length=width, padding="max_length", truncation=True)
            news.setdefault(doc_id, [tokens, cat, subcat])
            if mode == "train":
                _number(categories, cat)
                _number(subcategories, subcat)
    return (news, news_index, categories, subcategories) if mode == "train" else (news, news_index)


def get_doc_input_bert(news, news_index, category_dict, subcategory_dict, args):
    rows, width = len(news) + 1, args.num_words_title
    ids = np.zeros((rows, width), dtype="int32")
    attn = np.zeros((rows, width), dtype="int32")
    cats = np.zeros(rows, dtype="int32")
    subcats = np.zeros(rows, dtype="int32")
    for doc_id, (tokens, cat, subcat) in news.items():
        r = news_index[doc_id]
        ids[r], attn[r] = tokens["input_ids"], tokens["attention_mask"]
        cats[r], subcats[r] = category_dict.get(cat, 0), subcategory_dict.get(subcat, 0)
    return ids, attn, cats, subcats
