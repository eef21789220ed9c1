"""The learnable synthetic MIND-format corpus of the quality golden (tests/golden/quality_0.npz): pure numpy + the shared hash
generator, no reference code.  make_golden.golden_quality (build container: trains and evaluates the REFERENCE on it) and
tools/quality_proto.py (GPU box: the engine on it) both build it from here; the tests read the committed copy in the fixture."""
import numpy as np

import hashinit


def quality_corpus(seed=71, n_topics=12, per_topic=50, L=30, T=2, D=256, n_users=400, n_train=1600, n_test=600, U=50, npratio=4):
    """Synthetic MIND-format corpus with planted structure: `n_topics` topics x `per_topic` news, a title = 6 ... 20 tokens, 60 % from
    its topic's 30-token vocabulary and the rest from a 200-token common pool; a user prefers two topics (history: 85 % from them);
    teacher tables = a per-teacher topic centroid + noise (what a fine-tuned teacher's news vectors look like to the KD losses).
    Train impressions: 1 clicked news of a preferred topic + `npratio` unclicked ones (behaviors_np4 format, dataloader.py:118-150);
    test impressions: 5 ... 30 candidates with `-0` / `-1` labels (MIND dev format, dataloader.py:282-300), 10 % label noise so that
    no model reaches AUC 1.  Everything is a pure function of `seed` (hashinit generators)."""
    H = hashinit
    n_news = n_topics * per_topic
    topic = np.concatenate([[-1], np.repeat(np.arange(n_topics), per_topic)])
    topic_vocab = 1000 + np.arange(n_topics * 30).reshape(n_topics, 30)
    common = 5000 + np.arange(200)
    lens = H.hash_randint(seed, "q.len", (n_news + 1,), 6, 21)
    from_topic = H.hash_uniform(seed, "q.src", (n_news + 1, L), 0.0, 1.0) < 0.6
    pick_t = H.hash_randint(seed, "q.tt", (n_news + 1, L), 0, 30)
    pick_c = H.hash_randint(seed, "q.tc", (n_news + 1, L), 0, 200)
    comb = np.zeros((n_news + 1, 2 * L), np.int64)
    for i in range(1, n_news + 1):
        ids = np.where(from_topic[i], topic_vocab[topic[i]][pick_t[i]], common[pick_c[i]])
        comb[i, :lens[i]] = ids[:lens[i]]
        comb[i, L:L + lens[i]] = 1
    tables = []
    for t in range(T):
        cen = H.hash_normal(seed, "q.cen%d" % t, (n_topics, D), std=0.3)
        tab = H.hash_normal(seed, "q.tab%d" % t, (n_news + 1, D), std=0.12)
        tab[1:] += cen[topic[1:]]
        tables.append(tab.astype(np.float32))
    pref = np.stack([H.hash_randint(seed, "q.p0", (n_users,), 0, n_topics), H.hash_randint(seed, "q.p1", (n_users,), 0, n_topics)], 1)

    def draw_news(tag, n, user, p_pref):
        """n news ids for `user`: with probability p_pref from a preferred topic (the first one 2 : 1), else from any topic;
        p_pref < 0: from the topics the user does NOT prefer."""
        u = H.hash_uniform(seed, tag + ".u", (n,), 0.0, 1.0)
        second = H.hash_randint(seed, tag + ".w", (n,), 0, 3) == 2
        anyt = H.hash_randint(seed, tag + ".t", (n,), 0, n_topics)
        if p_pref < 0:
            others = np.array([t for t in range(n_topics) if t not in pref[user]])
            tp = others[anyt % len(others)]
        else:
            tp = np.where(u < p_pref, pref[user][second.astype(np.int64)], anyt)
        return 1 + tp * per_topic + H.hash_randint(seed, tag + ".i", (n,), 0, per_topic)

    def history(tag, user):
        n = int(H.hash_randint(seed, tag + ".hl", (1,), 3, U + 15)[0])          # some longer than user_log_length: truncated to the last U
        return draw_news(tag + ".h", n, user, 0.85)

    train_lines, test_lines = [], []
    for j in range(n_train):
        user = int(H.hash_randint(seed, "q.tru%d" % j, (1,), 0, n_users)[0])
        hist = history("q.tr%d" % j, user)
        pos = int(draw_news("q.trp%d" % j, 1, user, 1.0)[0])
        neg = draw_news("q.trn%d" % j, npratio, user, -1.0)
        train_lines.append("%d\tU%d\t11/15/2019 8:55:22 AM\t%s\tN%d\t%s" % (
            j, user, " ".join("N%d" % x for x in hist), pos, " ".join("N%d" % x for x in neg)))
    for j in range(n_test):
        user = int(H.hash_randint(seed, "q.teu%d" % j, (1,), 0, n_users)[0])
        hist = history("q.te%d" % j, user)
        nc = int(H.hash_randint(seed, "q.tec%d" % j, (1,), 5, 31)[0])
        cand = draw_news("q.ted%d" % j, nc, user, 0.3)
        noise = H.hash_uniform(seed, "q.tez%d" % j, (nc,), 0.0, 1.0) < 0.1
        lab = np.array([int(topic[x] in pref[user]) for x in cand]) ^ noise.astype(np.int64)
        if j % 97 == 5:
            lab[:] = 0                                                     # run.py:346 skips impressions without a click
        test_lines.append("%d\tU%d\t11/15/2019 8:55:22 AM\t%s\t%s" % (
            j, user, " ".join("N%d" % x for x in hist), " ".join("N%d-%d" % (x, l) for x, l in zip(cand, lab))))
    news_index = {"N%d" % i: i for i in range(1, n_news + 1)}
    return dict(news_combined=comb, tables=tables, news_index=news_index, train_lines=train_lines, test_lines=test_lines, topic=topic)


def quantize_delta(d):
    """Trained parameter - initial parameter as int8 with one fp32 scale per row (per tensor for vectors): the fixture carries
    14.7 M trained parameters, and 8 bits of a delta whose largest element is ~400 x lr are a perturbation of ~2 % of the
    delta's r.m.s. - the reference EVALUATES the dequantised weights, so what the engine is held to is exact either way."""
    d2 = d.reshape(d.shape[0], -1) if d.ndim > 1 else d.reshape(1, -1)
    scale = np.maximum(np.abs(d2).max(1, keepdims=True), 1e-30).astype(np.float32) / 127.0
    q = np.clip(np.rint(d2 / scale), -127, 127).astype(np.int8)
    return q.reshape(d.shape), scale.reshape(-1)


def dequantize_delta(q, scale):
    q2 = q.reshape(q.shape[0], -1) if q.ndim > 1 else q.reshape(1, -1)
    return (q2.astype(np.float32) * scale.reshape(-1, 1).astype(np.float32)).reshape(q.shape)
