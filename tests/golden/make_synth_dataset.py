#!/usr/bin/env python3
"""A seeded synthetic dataset in the ULTRA on-disk format (the reference's loader: ultra/utils/data_utils.py:99-180) with LEARNABLE
labels and a validation set large enough that an end-of-training NDCG@10 is a statement about the trainer, not about eight queries:

    python tests/golden/make_synth_dataset.py <out_dir>          (also imported: write_dataset(out_dir) -> {file: sha256})

train 500 / valid 400 / test 64 queries x 10 documents, 16 dense features in (-1, 1) with four decimals, relevance 0..4 = quantile
buckets of a fixed nonlinear score of the features + noise, the initial list = the documents ordered by a NOISY copy of that score
(a production ranker: clicks are position-biased towards good documents).  Pure numpy RandomState + fixed text formatting: the same
bytes everywhere, so the fixtures made from it (tests/golden/make_golden.py conv_dla_synth) pin batch checksums of it.
Data, not source: nothing of the reference is in here."""
import hashlib
import json
import os
import sys

import numpy as np

F, L = 16, 10
SPLITS = (("train", 500), ("valid", 400), ("test", 64))
SEED = 20260929


def write_dataset(out_dir, seed=SEED):
    rng = np.random.RandomState(seed)
    w1 = rng.normal(size=F)
    w2 = rng.normal(size=(F, 4))
    v2 = rng.normal(size=4)
    os.makedirs(out_dir, exist_ok=True)
    json.dump({"feature_size": F, "max_label": 4.0}, open(os.path.join(out_dir, "settings.json"), "w"))
    sums = {}
    # bucket edges from a large sample of the score: the same label distribution in every split
    probe = np.round(rng.uniform(-1, 1, size=(20000, F)), 4)

    def score(x):
        return x @ w1 + np.tanh(x @ w2) @ v2

    edges = np.quantile(score(probe) + 0.35 * rng.normal(size=probe.shape[0]), [0.45, 0.70, 0.85, 0.95])
    for split, nq in SPLITS:
        d = os.path.join(out_dir, split)
        os.makedirs(d, exist_ok=True)
        x = np.round(rng.uniform(-1, 1, size=(nq, L, F)), 4)
        s = score(x) + 0.35 * rng.normal(size=(nq, L))
        lab = np.searchsorted(edges, s).astype(np.int64)           # 0 .. 4
        order = np.argsort(-(s + 2.0 * rng.normal(size=(nq, L))), axis=1, kind="stable")  # the initial ranker
        feat_lines, list_lines, label_lines = [], [], []
        doc = 0
        for q in range(nq):
            ids = []
            for k in range(L):
                j = order[q, k]
                feat_lines.append("%s_%d_%d %s" % (split, q + 1, k + 1, " ".join("%d:%.4f" % (f + 1, x[q, j, f]) for f in range(F))))
                ids.append(doc)
                doc += 1
            list_lines.append("%d %s" % (q + 1, " ".join(str(i) for i in ids)))
            label_lines.append("%d %s" % (q + 1, " ".join("%.1f" % lab[q, order[q, k]] for k in range(L))))
        for suffix, lines in (("feature", feat_lines), ("init_list", list_lines), ("labels", label_lines)):
            path = os.path.join(d, "%s.%s" % (split, suffix))
            data = ("\n".join(lines) + "\n").encode("ascii")
            open(path, "wb").write(data)
            sums["%s/%s.%s" % (split, split, suffix)] = hashlib.sha256(data).hexdigest()
    return sums


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "./ultra_synth_data"
    for k, v in sorted(write_dataset(out).items()):
        print(v, k)
