"""Parity bookkeeping for the -m gpu tests: every comparison against a golden / the oracle goes
through `close()`, which asserts the north-star bound (BASELINE.json: "NMF/conv within 1e-4 fp32",
read as max|got − ref| ≤ rel · max|ref| + floor) and records the ACHIEVED maximum error, so the
run leaves `gpurun_out/parity.json` behind (tests/conftest.py writes it at session end; the copy
that is judged is committed as profiles/rNN_parity.json)."""
from __future__ import annotations

import os

import torch

RECORDS = []


def _test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def note(what, **kv):
    """Record a scalar fact of the run (e.g. the fraction of HALS matrices excluded by the gate margin)."""
    RECORDS.append({"test": _test_id(), "what": what, **kv})


def close(what, got, ref, rel=1e-4, floor=1e-6, why=None, extra=0.0):
    """assert max|got − ref| ≤ rel·max|ref| + floor (+ extra); record the achieved error.
    `why` must be given whenever rel > 1e-4: it is the stated reason for the looser bound."""
    assert rel <= 1e-4 or why, f"{what}: a bound looser than 1e-4 needs its reason"
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    if ref.numel() == 0:
        return 0.0
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    bound = rel * scale + floor + float(extra)
    RECORDS.append({"test": _test_id(), "what": what, "max_abs_err": err, "ref_max": scale,
                    "rel_err": err / (scale + 1e-30), "bound_rel": rel, "bound": bound,
                    **({"why": why} if why else {})})
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e} (rel {rel:g} of max|ref| {scale:.3e})" + (
        f" [{why}]" if why else "")
    return err
