"""Every path that ships behind a runtime knob under the parity bars, in the driver's own GPU test run (round 4
had this as a builder-run shell script, tools/knob_matrix.sh).  Per knob set: the reference's config-2 fixture (`ipw_cfg2`: one
IPW + DNN[256,256] step recorded from the reference, base_algorithm.py:208-226 / ipw_rank.py:102-182) through the ONE-call product
step, and a BASELINE-shaped full-size step against the oracle (config 3 = DLA at 10 240 rows, or config 4 = PairDebias at
12 800 x 700, whichever the knob touches).  The knobs are read when an engine is built (ultr_config_reload)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.hipref import dev, load_golden  # noqa: E402
from tests.test_gpu_full_size import CONFIGS, aux_for, make_inputs, run_oracle  # noqa: E402

# (knob settings, the full-size config that exercises them)
KNOB_SETS = [
    ("ULTR_WG_H3=2", "cfg3_dla"),                      # split-half weight gradients at any batch size (config 2 takes them too)
    ("ULTR_WG_H3=0", "cfg4_pairdebias"),               # register-tile weight gradients at every size
    ("ULTR_FWD_R=32", "cfg3_dla"),                     # 32-row forward tiles (fp32 matrix cores: the split-half stream is 16-row only)
    ("ULTR_BWD_R=32", "cfg3_dla"),                     # 32-row backward tiles
    ("ULTR_BIG_FWD=2 ULTR_BIG_BWD=2", "cfg3_dla"),     # the per-layer path wherever it is legal
    ("ULTR_BIG_FWD=0 ULTR_BIG_BWD=0", "cfg4_pairdebias"),  # ... and nowhere (row tiles at config 4)
    ("ULTR_NO_FUSED_FB=1", "cfg3_dla"),                # separate forward / loss / backward kernels at config 2's size
    ("ULTR_FB_MAX_WG_PER_CU=2", "cfg3_dla"),
    ("ULTR_FWD_NW=4 ULTR_BWD_NW=4", "cfg3_dla"),       # general-shape kernels with 4 waves
    ("ULTR_FWD_NW=16 ULTR_BWD_NW=16", "cfg3_dla"),
    ("ULTR_NO_VEC=1", "cfg3_dla"),                     # the unaligned-shape (scalar load) builds on aligned shapes
    ("ULTR_FWD_Q4=0", "cfg4_pairdebias"),
    ("ULTR_WGRAD_WGS=300", "cfg3_dla"),
    ("ULTR_FB_H3=0 ULTR_FWD_H3=0 ULTR_BWD_H3=0 ULTR_WG_H3=0", "cfg4_pairdebias"),  # every product on the fp32 matrix cores
    # round 5: configs 3 / 4 take the wide-tile kernels by default - the 16-row tiles (dnn_fwd_kernel, dnn_bwd2_kernel) and the
    # per-layer backward they replaced stay under the same bars
    ("ULTR_FWD_WIDE=0 ULTR_BWD_WIDE=0", "cfg3_dla"),
    ("ULTR_FWD_WIDE=0 ULTR_BWD_WIDE=0", "cfg4_pairdebias"),
    ("ULTR_FWD_WIDE=0", "cfg4_pairdebias"),            # 16-row forward in front of the wide backward (and the other way round below)
    ("ULTR_BWD_WIDE=0", "cfg3_dla"),
]


@pytest.fixture(params=KNOB_SETS, ids=[k[0].replace(" ", ",") for k in KNOB_SETS])
def knobs(request, monkeypatch):
    from ultra_pytorch_amd import _lib
    for kv in request.param[0].split():
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    _lib.load().ultr_config_reload()
    yield request.param
    monkeypatch.undo()
    _lib.load().ultr_config_reload()


def product_step(F, hidden, B, L, algo, lr, params, state, feats, ids, y, aux, ipw):
    from ultra_pytorch_amd import engine, hip_ops
    shape = hip_ops.DnnShape(F, hidden, "elu")
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo=algo, learning_rate=lr)
    p = dev(params)
    st = None if state is None else dev(state)
    a = None if aux is None else dev(aux)
    tab = None if ipw is None else dev(np.asarray(ipw, np.float32))
    eng.train_step(p, st, dev(np.asarray(feats, np.float32)), feats.shape[0], dev(ids, torch.int32), dev(y, torch.float32), aux=a, ipw_table=tab)
    sc = eng.read_scalars()
    n = shape.n_params
    tail = eng.grads[n:].cpu().numpy()
    gs = 1.0 if algo == "pairdebias" else 1.0 / tail[1]
    out = dict(scores=eng.scores.cpu().numpy(), loss=float(sc[0]), norm=float(sc[1]), grads=eng.grads[:n].cpu().numpy() * gs,
               params=p.cpu().numpy(), state=None if st is None else st.cpu().numpy())
    eng.close()
    return out


def test_reference_fixture_under_the_knob(knobs):
    d, m = load_golden("ipw_cfg2")
    r = product_step(m["F"], m["hidden"], m["B"], m["L"], "softmax", m["lr"], d["s0_pre_params"], d["s0_pre_adagrad"], d["s0_features"],
                     d["s0_docids"], d["s0_labels"], None, d["ipw_list"])
    np.testing.assert_allclose(r["scores"], d["s0_scores"], atol=1e-5, rtol=0)
    assert abs(r["loss"] - float(d["s0_loss"])) <= 1e-5 * max(1.0, abs(float(d["s0_loss"])))
    g = d["s0_grads"]
    np.testing.assert_allclose(r["grads"], g, rtol=1e-5, atol=1e-6 * float(np.abs(g).max()))
    assert abs(r["norm"] - float(d["s0_norm"])) <= 1e-5 * float(d["s0_norm"])
    sel = np.abs(g) > 1e-4 * np.abs(g).max()  # (Adagrad's first step is lr x sign(g): entries that are rounding noise excluded)
    np.testing.assert_allclose(r["params"][sel], d["s0_post_params"][sel], atol=2e-6, rtol=1e-5)
    # s' = s + g^2 with g at the 1e-5 bar: 2e-5 relative (+ 2e-6 x max for elements with g ~ 0), as tests/test_gpu_parity.py
    np.testing.assert_allclose(r["state"], d["s0_post_adagrad"], rtol=2e-5, atol=2e-6 * float(d["s0_post_adagrad"].max()))


_ORACLE = {}  # the oracle's step per config: the same for every knob (computed once, ~1 s each on the host)


def oracle_case(name):
    if name not in _ORACLE:
        from oracle import ultr_oracle as O
        F, hidden, B, L, algo, lr = CONFIGS[name]
        rng = np.random.RandomState(7)
        feats, ids, y = make_inputs(F, B, L, algo)
        params = O.init_params(F, hidden, seed=2)
        aux = aux_for(name, rng)
        _ORACLE[name] = (feats, ids, y, params, aux, run_oracle(name, params, feats, ids, y, aux))
    return _ORACLE[name]


def test_full_size_step_under_the_knob(knobs):
    from ultra_pytorch_amd import synthetic
    name = knobs[1]
    F, hidden, B, L, algo, lr = CONFIGS[name]
    feats, ids, y, params, aux, ref = oracle_case(name)
    r = product_step(F, hidden, B, L, algo, lr, params, None if algo == "dla" else np.zeros_like(params), feats, ids, y, aux,
                     synthetic.load_ipw() if algo == "softmax" else None)
    np.testing.assert_allclose(r["scores"], ref["scores"], atol=1e-5)
    assert abs(r["loss"] - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
    assert abs(r["norm"] - ref["norm"]) <= 1e-5 * ref["norm"]
    gref = ref["grads"]
    np.testing.assert_allclose(r["grads"], gref, rtol=1e-5, atol=1e-5 * float(np.abs(gref).max()))
