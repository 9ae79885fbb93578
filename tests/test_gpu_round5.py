"""GPU evidence added in round 5: closures that only LOOK affine keep the callback path's result, the unguided full
batch against the C++ port, and the fp16-pair node GEMMs held to the fp32 instruction's accuracy."""
import os
import types

import numpy as np
import pytest

from tests.helpers import TINY, TINY_P, cfg_of, edm_from_cfg, pred_from_cfg

pytestmark = pytest.mark.gpu


def test_closures_that_only_look_affine_follow_the_callback_path(golden):
    """a target closure whose python control flow depends on t -- here guidance with twice the weight
    inside a window of t that the three-point affine probe does not visit -- is recognised as affine at the probed points, runs
    fused, and is caught by the all-t check that the host runs beside the device call: the fused result is discarded and the
    same call (same noise stream) runs through the callback path.  Result = PredTarget on the same closure, bit for bit; a
    closure that is switched OFF at the probed points (w = 0) never reaches the fused kernel at all."""
    import torch

    from gaudi_amd import sampling_edm
    from gaudi_amd.models_edm import PredTarget, get_cond_predictor_model, get_model
    g = golden("g7_end_to_end")
    cfg = cfg_of(g, "hetro_tiny")
    base = dict(dataset=cfg["dataset"], amp=cfg["amp"])
    eargs, esd = edm_from_cfg(dict(base, over=TINY, wseed=cfg["eseed"]), diffusion_steps=cfg["T"])
    pargs, psd = pred_from_cfg(dict(base, over=TINY_P, wseed=cfg["pseed"]))
    model, _, _ = get_model(eargs, state_dict=esd)
    cond_predictor = get_cond_predictor_model(pargs, None, state_dict=psd)
    nodes = [3, 5, 4, 2]
    args = types.SimpleNamespace(device="cuda", dataset=cfg["dataset"], max_nodes=10)
    T = cfg["T"]

    def in_window(t):
        t = float(torch.as_tensor(t).reshape(-1)[0])
        return 0.6 < t < 0.9

    def windowed(_input, _node_mask, _edge_mask, _t):
        pred = cond_predictor(_input, _node_mask, _edge_mask, _t)
        return -2 * pred[:, 1] if in_window(_t) else -pred[:, 1]

    def off_at_the_probes(_input, _node_mask, _edge_mask, _t):
        pred = cond_predictor(_input, _node_mask, _edge_mask, _t)
        return -pred[:, 1] if in_window(_t) else 0 * pred[:, 1]

    def run(target):
        model.seed, model.sample_offset = 11, 0
        model.engine.profile_reset(True)
        x, h, _, _ = sampling_edm.sample_guidance(args, model, target, nodes, scale=0.6)
        n = model.engine.profile_get()[0]
        model.engine.profile_reset(False)
        return x.numpy(), h.numpy(), n, dict(model.last_diag)

    fused = -(-T // 25)
    for closure in (windowed, off_at_the_probes):
        fn = lambda p, t, c=closure: _through(cond_predictor, c, p, t)
        xa, ha, na, da = run(closure)
        xb, hb, nb, _ = run(PredTarget(cond_predictor, fn))
        assert np.array_equal(xa, xb) and np.array_equal(ha, hb)
        if closure is windowed:  # fused chain (discarded) + the callback chain
            assert da.get("affine_recheck_failed") == 1 and na == fused + nb, (na, nb)
        else:
            assert "affine_recheck_failed" not in da and na == nb, (na, nb)
    # the guidance inside the window matters (otherwise the test would pass on an unguided chain)
    x0, _, _, _ = run(lambda z, nm, em, t: -cond_predictor(z, nm, em, t)[:, 1])
    assert not np.array_equal(x0, xa)
    model.engine.close()


def _through(cond_predictor, closure, p, t):
    """closure(z, nm, em, t) with the attached predictor returning p (what GaudiModel._trace_closure does)."""
    import torch
    cond_predictor._override, cond_predictor._override_used = p, False
    try:
        return closure(None, None, None, torch.full((p.shape[0], 1), float(t)))
    finally:
        cond_predictor._override = None
