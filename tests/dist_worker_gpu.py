"""Worker for tests/test_gpu_round2.py::test_two_ranks_real_engine: launched by torch.distributed.run (gloo) with both
ranks on GPU 0.  Each rank runs the REAL Engine on its contiguous shard of the global batch (global-index-keyed Philox
noise, global N), then ONE all_gather; rank 0 also runs the unsharded batch for the bit-for-bit comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from gaudi_amd import dist as gdist  # noqa: E402
from gaudi_amd import synth  # noqa: E402
from gaudi_amd.engine import Engine  # noqa: E402
from gaudi_amd.sampling_edm import build_masks  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    T, seed = 40, 77
    eargs = synth.edm_args(nf=64, n_layers=3, diffusion_steps=T)
    pargs = synth.pred_args(nf=60, n_layers=3)  # 60 -> padded to 64: the fused (64, 64) instantiation
    eng = Engine(0)
    eng.load_edm(eargs, synth.synth_edm_state_dict(eargs, 1, seed=11))
    eng.load_predictor(pargs, synth.synth_predictor_state_dict(pargs, 1, 5, seed=12))
    eng.set_steps_per_launch(7)  # 40 steps = 5 full launches + one of 5
    nodes = np.array([5, 7, 3, 7, 6, 11, 2, 9, 11, 4, 8])  # global batch of 11 -> shards of 6 and 5, padded to N = 11
    nm3, em_flat, _ = build_masks(nodes, int(nodes.max()), False)
    B, N = nm3.shape[0], nm3.shape[1]
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    w = np.array([0, -1, 0, 0, 0], np.float32)

    def sample_fn(nm_s, em_s, offset):
        x, h, _ = eng.sample(nm_s, em_s, seed=seed, sample_offset=offset, target_w=w, scale=0.6)
        return x, h

    lo, hi, x, h = gdist.sample_sharded(sample_fn, nm, em, rank, world)
    xs, hs = gdist.gather_to_all(x, h, B, N, 1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=xs, h=hs, lo=lo, hi=hi)
    if rank == 0:
        x_full, h_full = sample_fn(nm, em, 0)
        np.savez(os.path.join(out_dir, "unsharded.npz"), x=x_full, h=h_full)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        with open(os.path.join(sys.argv[1], f"err{os.environ.get('RANK', '0')}.txt"), "w") as f:
            f.write(traceback.format_exc())
        raise
