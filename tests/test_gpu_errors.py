"""Error behaviour of the C ABI on the GPU box: a bad argument or an unsupported shape comes back as a negative
ULTR_E_* code BEFORE anything is launched (nothing throws, nothing is written), and the Python layer turns it into an
exception - the counterpart of the reference raising on a malformed feed (click_simulation_feed.py:118-120)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

E_BADARG, E_UNSUPPORTED = -1, -2


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def test_bad_arguments_are_rejected_without_side_effects():
    from ultra_pytorch_amd import _lib, hip_ops
    lib = _lib.load()
    dev = torch.device("cuda")
    B, L = 4, 6
    scores = torch.full((B, L), 7.0, device=dev)
    labels = torch.ones(L, B, device=dev)
    ds = torch.full((B, L), -3.0, device=dev)
    ws = torch.zeros(hip_ops.loss_workspace_bytes(B, L) // 4, device=dev)
    null = ctypes.c_void_p(None)
    # NULL pointers, zero / negative sizes
    assert lib.ultr_softmax_ce(null, _vp(labels), null, null, 0, B, L, _vp(ds), _vp(ws), null) == E_BADARG
    assert lib.ultr_softmax_ce(_vp(scores), _vp(labels), null, null, 0, 0, L, _vp(ds), _vp(ws), null) == E_BADARG
    assert lib.ultr_softmax_ce(_vp(scores), _vp(labels), null, null, 0, B, -1, _vp(ds), _vp(ws), null) == E_BADARG
    # an IPW table pointer with a non-positive length
    tab = torch.ones(4, device=dev)
    assert lib.ultr_softmax_ce(_vp(scores), _vp(labels), null, _vp(tab), 0, B, L, _vp(ds), _vp(ws), null) == E_BADARG
    assert lib.ultr_dla_loss(_vp(scores), _vp(labels), null, 0, B, L, _vp(ds), _vp(ws), null) == E_BADARG
    assert lib.ultr_pairdebias_loss(_vp(scores), _vp(labels), null, null, B, L, B, _vp(ds), _vp(ws), null) == E_BADARG
    assert lib.ultr_lambdarank_loss(_vp(scores), _vp(labels), null, null, 1.0, B, L, _vp(ds), _vp(ws), null) == E_BADARG
    torch.cuda.synchronize()
    assert float(ds.min()) == -3.0 and float(ds.max()) == -3.0  # nothing ran
    # descriptors
    bad = _lib.DnnDesc()
    bad.feature_size, bad.n_hidden = 0, 1
    assert lib.ultr_dnn_param_count(ctypes.byref(bad)) <= 0
    bad.feature_size, bad.n_hidden = 8, 99
    assert lib.ultr_dnn_param_count(ctypes.byref(bad)) <= 0
    sr = _lib.SetRankDesc()
    sr.feature_size, sr.d_model, sr.num_heads, sr.num_layers, sr.dff = 8, 30, 4, 1, 8  # d_model % heads != 0
    assert lib.ultr_setrank_param_count(ctypes.byref(sr)) <= 0
    assert lib.ultr_loss_part_count(0) == 0 and lib.ultr_loss_workspace_bytes(0, 5) == 0


def test_unsupported_shapes_raise_in_python():
    from ultra_pytorch_amd import _lib, engine, hip_ops
    dev = torch.device("cuda")
    # a list that does not fit the list-wise kernels' LDS staging: the loss stage reports it, the engine raises
    B, L = 1, 20000
    for algo in ("softmax", "pairdebias"):
        eng = engine.StepEngine(hip_ops.DnnShape(8, [4]), B, L, dev, algo=algo)
        aux = torch.ones(2 * L, device=dev) if algo == "pairdebias" else None
        with pytest.raises(_lib.UltrHipError):
            eng.loss(torch.ones(L, B, device=dev), aux=aux)
    # SetRank training past the attention cap
    shape = hip_ops.SetRankShape(8, 32, 2, 1, 8)
    with pytest.raises((_lib.UltrHipError, ValueError, NotImplementedError)):
        e2 = engine.SetRankStepEngine(shape, 1, 300, dev, algo="softmax")
        f = torch.zeros(300, 8, device=dev)
        ids = torch.arange(300, dtype=torch.int32, device=dev).view(300, 1)
        p = torch.zeros(shape.n_params, device=dev)
        e2.forward(p, f, 300, ids, train=True)
        e2.loss(torch.ones(300, 1, device=dev))
        e2.backward(p, f, 300, ids)
        torch.cuda.synchronize()
