"""CPU-only checks: the C-ABI library loads and exports every symbol include/ultr_hip.h declares (no compute
calls), host-side sizing queries, the hparam grammar, the plugin seam, host metrics vs the oracle, and that the
product refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ultr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ultr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ultra_pytorch_amd import _lib, build
    build.build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libultr_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree: %s" % (set(names) ^ set(_lib.SIGNATURES))
    assert _lib.load().ultr_abi_version() == _lib.ABI_VERSION == 8


def test_header_is_plain_c(tmp_path):
    """include/ultr_hip.h is the drop-in boundary: it must compile as C99 (no torch / C++ types in any signature) and as C++."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "include/ultr_hip.h"\nint main(void) { ultr_step_args a; ultr_setrank_desc s; (void)a; (void)s; return ULTR_ABI_VERSION; }\n')
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", root, str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", root, str(src)])


def test_host_only_queries_match_reference_layout():
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import hip_ops
    for F, hidden in ((136, [256, 256]), (700, [512, 256, 128]), (13, [19, 6, 3]), (136, [])):
        shape = hip_ops.DnnShape(F, hidden)
        assert shape.n_params == O.num_params(F, hidden)
        assert [(n, tuple(s), o) for n, s, o in shape.layout()] == [(n, tuple(s), o) for n, s, o in O.param_layout(F, hidden)]
        assert shape.saved_bytes(2560) > 0 and shape.bwd_workspace_bytes(2560) > 0
    assert hip_ops.tail_floats(10) == 24
    assert hip_ops.loss_workspace_bytes(256, 10) >= 64 * 24 * 4
    with pytest.raises(ValueError):
        hip_ops.DnnShape(0, [4])


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ultra_pytorch_amd import engine, hip_ops
    from ultra_pytorch_amd.ranking_model import DNN
    with pytest.raises(RuntimeError):
        engine.StepEngine(hip_ops.DnnShape(8, [4]), 2, 3, torch.device("cpu"))
    with pytest.raises(RuntimeError):
        DNN("hidden_layer_sizes=[4]", 8).build([torch.zeros(2, 8)])
    from ultra_pytorch_amd.learning_algorithm import IPWrank

    class DS:
        feature_size = 8
    with pytest.raises(RuntimeError):
        IPWrank(DS(), {"learning_algorithm_hparams": "", "ranking_model": "ultra_pytorch_amd.ranking_model.DNN",
                       "ranking_model_hparams": "", "max_candidate_num": 3, "selection_bias_cutoff": 3})


def test_hparams_grammar():
    from ultra_pytorch_amd.utils import HParams
    h = HParams(hidden_layer_sizes=[512, 256, 128], activation_func="elu", learning_rate=0.05, flag=False, n=3)
    h.parse("hidden_layer_sizes=[256, 256],learning_rate=0.01,unknown_key=3,flag=true,n=7")
    assert h.hidden_layer_sizes == [256, 256] and h.learning_rate == 0.01 and h.flag is True and h.n == 7
    h.parse("hidden_layer_sizes[1]=64,activation_func=relu")
    assert h.hidden_layer_sizes == [256, 64] and h.activation_func == "relu"
    assert HParams(a=1).parse("").a == 1
    with pytest.raises(ValueError):
        HParams(a=1).parse("a=1,a=2")
    with pytest.raises(ValueError):
        HParams(a=1).parse("a=x")
    with pytest.raises(ValueError):
        HParams(a=[1]).parse("a=3")


def test_plugin_seam():
    from ultra_pytorch_amd.utils import find_class
    assert find_class("ultra_pytorch_amd.learning_algorithm.DLA").__name__ == "DLA"
    assert find_class("ultra_pytorch_amd.ranking_model.Linear").__name__ == "Linear"
    with pytest.raises(ImportError):
        find_class("ultra_pytorch_amd.learning_algorithm.Nope")


def test_state_dict_keys_and_roundtrip():
    from ultra_pytorch_amd.ranking_model import DNN
    m = DNN("hidden_layer_sizes=[8,4]", 6)
    keys = list(m.state_dict().keys())
    assert keys == ["sequential.layer_norm%d.%s" % (j, k) if "ln" in t else "sequential.linear%d.%s" % (j, k)
                    for j in range(3) for t, k in (("ln", "weight"), ("ln", "bias"), ("lin", "weight"), ("lin", "bias"))]
    m2 = DNN("hidden_layer_sizes=[8,4]", 6)
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m.flat_params, m2.flat_params)
    # parameters are views of the flat vector: an in-place edit of the flat buffer shows through the module
    m.flat_params.zero_()
    assert float(m.state_dict()["sequential.linear0.weight"].abs().sum()) == 0.0


def test_host_metrics_match_oracle():
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd.utils import metrics as M
    rng = np.random.RandomState(0)
    y = torch.tensor(rng.randint(0, 5, size=(7, 9)).astype(np.float32))
    s = torch.tensor(rng.normal(size=(7, 9)).astype(np.float32))
    topn = [1, 3, 5, 10]
    M.RankingMetricKey.MAX_LABEL = 4.0
    np.testing.assert_allclose(M.make_ranking_metric_fn("ndcg", topn)(y, s, None), O.ndcg(y, s, topn), atol=1e-6)
    np.testing.assert_allclose(M.make_ranking_metric_fn("mrr", topn)(y, s, None), O.mrr(y, s, topn), atol=1e-6)
    np.testing.assert_allclose(M.make_ranking_metric_fn("err", topn)(y, s, None), O.err(y, s, topn, 4.0), atol=1e-6)


def test_synthetic_batch_contract():
    from ultra_pytorch_amd import synthetic
    feats, ids, clicks = synthetic.make_batch(np.random.RandomState(0), 16, 10, 136, n_pad=2)
    assert feats.shape == (16 * 8, 136) and ids.shape == (10, 16) and clicks.shape == (10, 16)
    assert (ids[8:] == feats.shape[0]).all() and (clicks.sum(0) > 0).all() and (clicks[8:] == 0).all()
    assert len(synthetic.load_ipw()) == 40


def test_ranklist_by_scores_follows_reference_semantics():
    """data_utils.py:567-617: positions are ranked (not document ids), pads (doc index < 0) are dropped, and a score
    matrix whose shape does not match the initial lists raises instead of producing a truncated run file."""
    from ultra_pytorch_amd.utils import data_utils

    class D:
        qids = ["q1", "q2"]
        dids = ["a", "b", "c", "d"]
        initial_list = [[0, 1, 1, -1], [2, 3, -1, -1]]  # the same document twice in q1: both positions are kept

    out = data_utils.generate_ranklist_by_scores(D, [[0.1, 0.9, 0.5, 7.0], [1.0, 2.0, 3.0, 4.0]])
    assert out["q1"] == [("b", 0.9), ("b", 0.5), ("a", 0.1)]
    assert out["q2"] == [("d", 2.0), ("c", 1.0)]
    with pytest.raises(ValueError):
        data_utils.generate_ranklist_by_scores(D, [[0.1, 0.9, 0.5, 7.0]])
    with pytest.raises(ValueError):
        data_utils.generate_ranklist_by_scores(D, [[0.1, 0.9, 0.5], [1.0, 2.0, 3.0, 4.0]])


def test_setrank_forward_consumes_the_reference_random_stream():
    """SetRank.py:245-246 shuffles an index list on every forward: the same draws must be consumed here."""
    import random
    from ultra_pytorch_amd import engine
    random.seed(5)
    engine._setrank_draw(10)
    got = random.random()
    random.seed(5)
    ind = list(range(10))
    random.shuffle(ind)
    assert got == random.random()


def test_library_has_no_packed_fp32_instructions():
    """gfx950 hazard (profiles/r04_h3_rootcause.md): v_pk_{mul,add}_f32 with op_sel:[0,1] reads its operand as zero in lanes
    48..63 while the SIMD's other wave executes f16 MFMAs.  The library is built with packed fp32 selection off; this
    disassembles the gfx950 code objects of the binary that ships and fails on any packed fp32 instruction."""
    from ultra_pytorch_amd import build
    assert len(build.device_code_objects(build.LIB)) >= 6  # one per kernel translation unit
    found = build.audit_isa(build.LIB)
    if found is None:
        pytest.skip("no llvm-objdump on this box")
    assert found == [], found[:5]


def test_graft_entry_build_passes():
    """The driver's build check (round 4: its hard-coded ABI number went stale when the header moved on)."""
    import __graft_entry__ as g
    g.build()
