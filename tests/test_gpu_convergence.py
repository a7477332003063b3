"""End-of-training NDCG@10 against the reference's own main.py for the click-feed algorithms (SURVEY 8c last sentence,
BASELINE's "NDCG@10 parity"): tests/golden/conv_{ipw,dla,pairdebias}.npz hold, per seed, the reference driver's run on the toy
ULTRA dataset (ClickSimulationFeed + PBM, DNN[32,16], batch 64, 350 steps, a checkpoint every 50) in three variants - 1 thread,
8 threads, initial weights moved by one fp32 rounding - i.e. the reference's own sensitivity to summation order.  The
counterpart driver (ultra_pytorch_amd.main) runs the same seeds / settings from the same initial weights and must
  (a) draw the identical batch sequence (the feeds share Python's `random` stream bit for bit),
  (b) end every seed at a validation NDCG@10 inside the band the reference's variants span (+ the band's own width, + one
      tie-flip quantum of this 8-query validation set),
  (c) agree on the mean over seeds within 0.005 (+ the distance between the reference variants' own means, which is 0 for IPW
      and PairDebias and 0.037 for DLA, whose stateless sign-like updates turn one rounding into a different trajectory).
Reference: main.py:85-227, ipw_rank.py:102-182, dla.py:179-266, pairwise_debias.py:106-174."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA = os.path.join(GOLDEN, "ultra_toy_data") + "/"


def feed_checksum(algo, input_feed, L):
    w = np.arange(1, L + 1, dtype=np.float64)[:, None]
    ids = np.stack([np.asarray(input_feed[algo.docid_inputs_name[l]], np.float64) for l in range(L)])
    lab = np.stack([np.asarray(input_feed[algo.labels_name[l]], np.float64) for l in range(L)])
    col = np.arange(1, ids.shape[1] + 1, dtype=np.float64)[None, :]
    return float((ids * w * col).sum()), float((lab * w * col).sum())


def run_seed(d, m, seed, tmp_path, monkeypatch, data_dir=None):
    from ultra_pytorch_amd import learning_algorithm as LA
    from ultra_pytorch_amd import main as driver
    cls = getattr(LA, m["class"])
    s = dict(m["settings"])
    for k in ("train_input_feed", "valid_input_feed", "test_input_feed", "ranking_model", "learning_algorithm"):
        s[k] = s[k].replace("ultra.", "ultra_pytorch_amd.", 1)
    work = os.path.join(str(tmp_path), "s%d" % seed)
    model_dir = work + "/model/"
    os.makedirs(model_dir)
    sf = os.path.join(work, "settings.json")
    json.dump(s, open(sf, "w"))
    init = {k: torch.from_numpy(d["s%d_init_%s" % (seed, k)].copy()) for k in m["param_keys"]}
    torch.save(init, os.path.join(model_dir, "%s.ckpt" % s["learning_algorithm"]))  # create_model loads the reference's initial weights
    rec = {"first": True, "losses": [], "sums": []}
    orig_train = cls.train

    def train(self, feed):
        if rec["first"]:
            rec["first"] = False
            if m["prop_keys"]:  # DLA's DenoisingNet is not in the checkpoint: start it where the reference's started
                # (the reference registers the Linear twice: linear_layer.* and propensity_net.0.* are the same two tensors, dla.py:31-37)
                self.propensity_model.load_state_dict({k: torch.from_numpy(d["s%d_prop_%s" % (seed, k)].copy())
                                                       for k in m["prop_keys"] if k.startswith("linear_layer.")})
        rec["sums"].append(feed_checksum(self, feed, self.exp_settings["selection_bias_cutoff"]))
        out = orig_train(self, feed)
        rec["losses"].append(float(out[0]))
        return out

    monkeypatch.setattr(cls, "train", train)
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    argv = ["--data_dir", data_dir or DATA, "--setting_file", sf, "--model_dir", model_dir, "--output_dir", work + "/out/"] + m["argv"]
    _, history = driver.main(argv)
    monkeypatch.setattr(cls, "train", orig_train)
    return rec, history


@pytest.mark.parametrize("name", ["conv_ipw", "conv_dla", "conv_pairdebias"])
def test_end_of_training_ndcg_matches_the_reference(name, tmp_path, monkeypatch, capsys):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = json.loads(str(d["meta"]))
    topn = m["topn"]
    i10 = topn.index(10)
    finals, ref_finals, report = [], [], []
    var_finals = []
    for seed in m["seeds"]:
        rec, history = run_seed(d, m, seed, tmp_path, monkeypatch)
        run = m["runs"][str(seed)]
        # stop rule and checkpoint schedule are the driver's (exact)
        assert len(rec["losses"]) == run["n_steps"]
        assert [h[0] for h in history] == run["ckpt_steps"]
        # (a) the identical batch sequence: which documents at which positions, which of them clicked - every step
        np.testing.assert_array_equal(np.asarray(rec["sums"]), d["s%d_feed_sums" % seed])
        # the first step is teacher-forced (same weights, same batch): loss to 1e-5; later steps follow the fp32 trajectory
        ref_losses = d["s%d_losses" % seed]
        assert abs(rec["losses"][0] - ref_losses[0]) <= 1e-5 * max(1.0, abs(ref_losses[0]))
        assert np.all(np.isfinite(rec["losses"]))
        ours = np.asarray([[h[2]["ndcg_%d" % n] for n in topn] for h in history])
        variants = np.stack([d["s%d_%s_ndcg" % (seed, v)] for v in m["variants"]])  # [variant, checkpoint, topn]
        lo, hi = variants[:, -1, i10].min(), variants[:, -1, i10].max()
        finals.append(ours[-1, i10])
        ref_finals.append(variants[0, -1, i10])
        var_finals.append(variants[:, -1, i10])
        report.append((seed, float(ours[-1, i10]), float(lo), float(hi), float(np.abs(ours[:, i10] - variants[0, :, i10]).max())))
    with capsys.disabled():
        print("\n%s: seed, final NDCG@10 here, reference band [lo, hi], max |diff| over the checkpoints" % name)
        for r in report:
            print("   seed %d  %.6f  [%.6f, %.6f]  %.2e" % r)
        print("   mean here %.6f, reference %.6f" % (np.mean(finals), np.mean(ref_finals)))
    quantum = float(d["tie_quantum"]) if "tie_quantum" in d.files else 0.0
    for seed, v, lo, hi, _ in report:
        band = (hi - lo) + quantum + 1e-6
        assert lo - band <= v <= hi + band, (seed, v, lo, hi)                      # (b)
    # (c) the mean over seeds: within 0.005 of the reference's - widened by how far the reference's OWN variants' means lie apart
    # (0 for IPW / PairDebias; DLA's sign-SGD updates (dla.py:141-177: fresh Adagrad every step) amplify one rounding into different
    # trajectories - its three variants' means span 0.716 .. 0.753 on this 8-query validation set)
    vmeans = np.mean(np.stack(var_finals), axis=0)
    mlo, mhi = float(vmeans.min()), float(vmeans.max())
    assert mlo - (mhi - mlo) - 0.005 <= np.mean(finals) <= mhi + (mhi - mlo) + 0.005, (np.mean(finals), mlo, mhi)


def test_dla_end_of_training_on_the_synthetic_dataset(tmp_path, monkeypatch, capsys):
    """DLA (dla.py:141-177, 179-266) end to end on a dataset where the figure can tell a right trainer from a wrong one (VERDICT r05
    item 5): tests/golden/make_synth_dataset.py - 400 validation queries, learnable labels - regenerated here byte for byte (sha256 of
    every file is in the fixture), 10 seeds x 350 steps of the counterpart driver from the reference's initial weights, against
    tests/golden/conv_dla_synth.npz = the reference's own main.py in eight variants per seed (1 / 8 threads, six one-rounding
    perturbations of the initial weights).  The reference's variants end 0.004 .. 0.018 apart per seed around NDCG@10 = 0.948:
      (a) identical batch sequence (checksums of every batch), the first step's loss to 1e-5;
      (b) per seed: the final NDCG@10 within 3.5 POOLED standard deviations of the seed's variant mean (pooled over the ten seeds'
          deviations from their own means: 0.0039 - a seed's eight variants alone give a noisy scale, 0.0021 .. 0.0058), and at least
          six of the ten seeds inside their own variants' [min, max] without any widening;
      (c) the mean over seeds within 0.004 of the mean of the variant means (3.3 pooled deviations of a ten-seed mean; the variant
          means themselves span 0.9467 .. 0.9497).
    Why not "every seed inside [min, max] +- one std of its variants" (the form VERDICT r05 sketched): a NINTH run of the reference
    itself leaves the [min, max] of eight exchangeable runs with probability 2/9, and the widened band with roughly one in ten - over
    ten seeds the reference would fail its own test about every second time.  Measured here (round 6, MI355X): nine of ten seeds
    inside the unwidened band, seed 1 at 0.9363 against [0.9429, 0.9500] (2.5 pooled deviations), mean 0.9477 against 0.9484.
    The old toy-set test accepted any mean in [0.64, 0.87]."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_synth_dataset as MS
    d = np.load(os.path.join(GOLDEN, "conv_dla_synth.npz"))
    m = json.loads(str(d["meta"]))
    data_dir = os.path.join(str(tmp_path), "synth") + "/"
    assert MS.write_dataset(data_dir, seed=m["dataset"]["seed"]) == m["dataset"]["sha256"], "the synthetic dataset is not the fixture's"
    topn = m["topn"]
    i10 = topn.index(10)
    finals, report, var_finals = [], [], []
    for seed in m["seeds"]:
        rec, history = run_seed(d, m, seed, tmp_path, monkeypatch, data_dir=data_dir)
        run = m["runs"][str(seed)]
        assert len(rec["losses"]) == run["n_steps"] and [h[0] for h in history] == run["ckpt_steps"]
        np.testing.assert_array_equal(np.asarray(rec["sums"]), d["s%d_feed_sums" % seed])          # (a)
        ref_losses = d["s%d_losses" % seed]
        assert abs(rec["losses"][0] - ref_losses[0]) <= 1e-5 * max(1.0, abs(ref_losses[0]))
        assert np.all(np.isfinite(rec["losses"]))
        ours = np.asarray([[h[2]["ndcg_%d" % n] for n in topn] for h in history])
        variants = np.stack([d["s%d_%s_ndcg" % (seed, v)] for v in m["variants"]])[:, -1, i10]
        lo, hi, sd = float(variants.min()), float(variants.max()), float(variants.std())
        finals.append(float(ours[-1, i10]))
        var_finals.append(variants)
        report.append((seed, finals[-1], lo, hi, sd, float(ours[0, i10])))
    vmeans = np.mean(np.stack(var_finals), axis=0)
    with capsys.disabled():
        print("\nconv_dla_synth: seed, final NDCG@10 here, the reference variants' [min, max], their std, NDCG@10 at the first checkpoint")
        for r in report:
            print("   seed %d  %.4f  [%.4f, %.4f]  %.4f   (%.4f)" % r)
        print("   mean over seeds here %.4f; the reference variants' means %.4f .. %.4f (mean %.4f)" % (np.mean(finals), vmeans.min(), vmeans.max(), vmeans.mean()))
    dev = np.stack(var_finals) - np.stack(var_finals).mean(axis=1, keepdims=True)
    pooled = float(np.sqrt((dev ** 2).sum() / (dev.size - dev.shape[0])))
    inside = 0
    for (seed, v, lo, hi, sd, _), vf in zip(report, var_finals):
        assert abs(v - float(vf.mean())) <= 3.5 * pooled, (seed, v, float(vf.mean()), pooled)           # (b)
        inside += int(lo <= v <= hi)
    assert inside >= 6, (inside, report)
    assert abs(np.mean(finals) - float(vmeans.mean())) <= 0.004, (np.mean(finals), float(vmeans.mean()), pooled)  # (c)
