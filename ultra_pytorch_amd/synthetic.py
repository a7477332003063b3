"""Seeded synthetic MSLR-style data: features U(-1,1) f32, labels U{0..4} with a relevant
document at position 0, PBM clicks (exam_prob x click_prob, last exam value reused beyond rank 10), lists with
no click rejected (as ClickSimulationFeed does, click_simulation_feed.py:89-91).  Host-side numpy only."""
import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
PBM_JSON = os.path.join(_DATA, "pbm_0.1_1.0_4_1.0.json")
IPW_JSON = os.path.join(_DATA, "randomized_pbm_0.1_1.0_4_1.0.json")


def load_pbm(path=PBM_JSON):
    d = json.load(open(path))
    return np.asarray(d["exam_prob"], np.float64) ** float(d.get("eta", 1.0)), np.asarray(d["click_prob"], np.float64)


def load_ipw(path=IPW_JSON):
    return np.asarray(json.load(open(path))["IPW_list"], np.float64)


def make_batch(rng, B, L, F, clicks=True, n_pad=0):
    """One training batch in the reference feed's layout: features [n_docs,F] f32, docids [L,B] int32
    (position-major; the last n_pad positions of every list are PAD = n_docs), labels [L,B] f32."""
    live = L - n_pad
    n_docs = B * live
    feats = rng.uniform(-1.0, 1.0, size=(n_docs, F)).astype(np.float32)
    docids = np.full((L, B), n_docs, dtype=np.int32)
    docids[:live, :] = (np.arange(B, dtype=np.int32)[None, :] * live + np.arange(live, dtype=np.int32)[:, None])
    rel = rng.randint(0, 5, size=(L, B))
    rel[0, :] = np.maximum(rel[0, :], 1)
    rel[live:, :] = 0
    if not clicks:
        return feats, docids, rel.astype(np.float32)
    exam, cp = load_pbm()
    ex = np.asarray([exam[l] if l < len(exam) else exam[-1] for l in range(L)])[:, None]
    pclick = ex * cp[np.minimum(rel, len(cp) - 1)]
    pclick[live:, :] = 0.0
    out = np.zeros((L, B), np.float32)
    todo = np.ones(B, bool)
    while todo.any():  # resample lists without clicks (the feed rejects them)
        c = (rng.uniform(size=(L, B)) < pclick).astype(np.float32)
        ok = todo & (c.sum(0) > 0)
        out[:, ok] = c[:, ok]
        todo &= ~ok
    return feats, docids, out
