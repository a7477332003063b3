import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    from tests import margins
    margins.flush()


H3_KNOBS = ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3", "ULTR_WG_H3")  # (the last one: the weight gradients of large batches)


@pytest.fixture(params=["split_half", "fp32_mfma"])
def mfma_mode(request, monkeypatch):
    """Both arithmetic plans of the wide layers' products under the same parity bars: the default (three v_mfma_f32_16x16x32_f16 on
    split hi / lo fp16 operands) and every product on v_mfma_f32_16x16x4_f32 (ULTR_FB_H3=0 ULTR_FWD_H3=0 ULTR_BWD_H3=0 - what the
    engine falls back to when a weight leaves the split-half copies' range, and bench.py's fp32_mfma_products figure).  The
    knobs are read when an engine is built (ultr_config_reload), so tests build their engines AFTER taking this fixture."""
    if request.param == "fp32_mfma":
        for k in H3_KNOBS:
            monkeypatch.setenv(k, "0")
    else:
        for k in H3_KNOBS:
            monkeypatch.delenv(k, raising=False)
    yield request.param
    monkeypatch.undo()
    try:
        from ultra_pytorch_amd import _lib
        _lib.load().ultr_config_reload()
    except Exception:
        pass


@pytest.fixture(params=["slabs", "split_half"])
def wgrad_path(request, monkeypatch):
    """The weight-gradient paths: 64 x 64 register tiles x row splits + a reduction launch (the default below 4096 rows) and
    dnn_wgrad_h3_kernel - 128 x 128 LDS-staged blocks on the fp16 matrix cores with split operands, the default from 4096 rows,
    forced onto these small shapes with ULTR_WG_H3=2.  (mfma_mode "fp32_mfma" sets ULTR_WG_H3=0 and wins: the combination
    fp32_mfma x split_half runs the register kernel again.)"""
    if request.param == "split_half" and os.environ.get("ULTR_WG_H3") != "0":
        monkeypatch.setenv("ULTR_WG_H3", "2")
    yield request.param
    monkeypatch.undo()
    try:
        from ultra_pytorch_amd import _lib
        _lib.load().ultr_config_reload()
    except Exception:
        pass
