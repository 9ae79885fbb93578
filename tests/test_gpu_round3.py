"""GPU evidence added in round 3: the RCCL code path executed on the hardware at hand (one rank), shard invariance of
heterogeneous batches through the plan hint, and the kernel-level changes of the round."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gaudi_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_gather_world1(tmp_path):
    """init_process_group("nccl", world_size=1, device_id=cuda:0) + gaudi_amd.dist.gather_to_all(..., device=dev) on the real
    engine's output + the plan check + bench.py's MAX all_reduce: the collective code of the N > 1 path runs over RCCL and
    returns the ungathered arrays bit for bit.  The RCCL version is logged (and kept in gpurun_out/ when that exists)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker_nccl.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=900)
    err = (tmp_path / "err0.txt").read_text() if (tmp_path / "err0.txt").exists() else ""
    assert r.returncode == 0, err + r.stderr[-2000:]
    info = json.loads((tmp_path / "nccl_world1.json").read_text())
    print("RCCL:", info)
    assert info["backend"] == "nccl" and info["same_x"] and info["same_h"] and info["finite"]
    assert (info["lo"], info["hi"]) == (0, 7) and info["allreduce"] == 1.25
    assert info["plan"][0] in (4, 8) and len(info["rccl_version"]) >= 2
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_world1.json"), "w") as f:
            json.dump(info, f)


def test_hetero_shards_run_one_plan_and_match_unsharded():
    """the kernel family / edge-GEMM arithmetic of a call follow from batch-wide maxima, so a shard without
    the batch's largest molecule could pick another plan.  With the plan hint (dist.sample_sharded(engine=...)) every shard
    runs the whole batch's plan and the concatenated shards equal the unsharded run bit for bit; without it they need not."""
    from gaudi_amd import dist as gdist
    from gaudi_amd.engine import Engine
    from gaudi_amd.sampling_edm import build_masks
    T = 12
    eargs = synth.edm_args(dataset="hetro", diffusion_steps=T)  # default widths: the LDS plan depends on the slot count
    pargs = synth.pred_args(dataset="hetro")
    F = synth.num_node_features("hetro")
    eng = Engine(0)
    eng.load_edm(eargs, synth.synth_edm_state_dict(eargs, F, seed=21))
    eng.load_predictor(pargs, synth.synth_predictor_state_dict(pargs, F, 5, seed=22))
    rings = np.array([3, 4, 3, 5, 4, 3, 10, 9])  # the big molecules sit in the second shard only
    nm3, em_flat, _ = build_masks(rings, 10, True)
    B, N = nm3.shape[0], nm3.shape[1]
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    w = np.array([3, 0, 1, 1, 0], np.float32)

    def sample_fn(nm_s, em_s, offset):
        x, h, _ = eng.sample(nm_s, em_s, seed=9, sample_offset=offset, target_w=w, scale=0.6)
        return x, h

    x_full, h_full = sample_fn(nm, em, 0)
    plan_full = (eng.kernel_variant()[1], eng.edge_math()[1])
    parts, plans = [], []
    for rank in range(2):
        lo, hi, x, h = gdist.sample_sharded(sample_fn, nm, em, rank, 2, engine=eng)
        parts.append((x, h))
        plans.append((eng.kernel_variant()[1], eng.edge_math()[1]))
    assert plans == [plan_full, plan_full], (plans, plan_full)
    assert np.array_equal(np.concatenate([p[0] for p in parts]), x_full)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), h_full)
    # the hint is cleared afterwards: a small batch on its own may plan differently again
    sample_fn(nm[:4], em[:4], 0)
    hint_free = (eng.kernel_variant()[1], eng.edge_math()[1])
    print("plans: whole batch", plan_full, "first shard alone", hint_free)
    eng.close()
