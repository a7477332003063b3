"""Measured parity margins of the widened-tolerance GPU tests (the full-size configs and the ill-conditioned `_odd` fixtures).

Each such test records its measured max-difference figures here; at session end they are written to
gpurun_out/parity_margins.json (pulled back from the GPU box with the logs; the reviewed copy is committed as
profiles/r06_parity_margins.json).  check() asserts that a figure stays within 2x of the committed one, so drift INSIDE a
widened tolerance band (1e-5 * max|g|, rtol 1e-4) becomes a test failure instead of passing silently.  The kernels are
deterministic (fixed-order sums), so the figures repeat exactly on the same build."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMITTED = os.path.join(ROOT, "profiles", "r06_parity_margins.json")
OUT = os.path.join(ROOT, "gpurun_out", "parity_margins.json")
FLOOR = 2e-7  # differences below fp32 resolution of O(1) values are noise, not margins

_measured = {}
_committed = json.load(open(COMMITTED)) if os.path.exists(COMMITTED) else {}


def check(test, key, value):
    value = float(value)
    _measured.setdefault(test, {})[key] = value
    ref = _committed.get(test, {}).get(key)
    if ref is not None:
        assert value <= 2.0 * max(float(ref), FLOOR), \
            "%s / %s drifted: measured %.3e, committed %.3e (profiles/r06_parity_margins.json)" % (test, key, value, ref)


def flush():
    if not _measured:
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    old = {}
    if os.path.exists(OUT):
        try:
            old = json.load(open(OUT))
        except Exception:
            old = {}
    old.update(_measured)
    with open(OUT, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)
