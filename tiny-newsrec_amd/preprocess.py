"""news.tsv -> news_index / padded token tables (Tiny-NewsRec/preprocess.py:15-66), TF-free.

The tokenizer is the caller's (transformers BertTokenizer on the reference's vocab file by default): one-off
host preprocessing, out of scope for kernels.  Output layout is the boundary:
news_index = {doc_id: 1..n} in file order, row 0 = all-zero pad news, int32 (n+1, L) ids and masks."""
import numpy as np


def make_tokenizer(args):
    from transformers import BertTokenizer
    return BertTokenizer(vocab_file=args.tokenizer_name, do_lower_case=True)


def read_news_bert(news_path, args, mode="train", tokenizer=None):
    tok = tokenizer or make_tokenizer(args)
    news, news_index, category_dict, subcategory_dict = {}, {}, {}, {}
    L = args.num_words_title
    with open(news_path, "r", encoding="utf-8") as f:
        for line in f:
            doc_id, category, subcategory, title = line.strip("\n").split("\t")[:4]
            if doc_id not in news_index:
                news_index[doc_id] = len(news_index) + 1
            enc = tok(title.lower(), max_length=L, padding="max_length", truncation=True)
            if doc_id not in news:
                news[doc_id] = [enc, category, subcategory]
            if mode == "train":
                category_dict.setdefault(category, len(category_dict) + 1)
                subcategory_dict.setdefault(subcategory, len(subcategory_dict) + 1)
    if mode == "train":
        return news, news_index, category_dict, subcategory_dict
    if mode == "test":
        return news, news_index
    raise AssertionError("Wrong mode!")


def get_doc_input_bert(news, news_index, category_dict, subcategory_dict, args):
    n = len(news) + 1
    L = args.num_words_title
    title = np.zeros((n, L), dtype="int32")
    mask = np.zeros((n, L), dtype="int32")
    cat = np.zeros(n, dtype="int32")
    sub = np.zeros(n, dtype="int32")
    for key, (enc, c, s) in news.items():
        i = news_index[key]
        title[i] = enc["input_ids"]
        mask[i] = enc["attention_mask"]
        cat[i] = category_dict.get(c, 0)
        sub[i] = subcategory_dict.get(s, 0)
    return title, mask, cat, sub
