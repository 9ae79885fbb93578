"""Worker for tests/test_dist_cpu.py: launched by torch.distributed.run with backend gloo.  Exercises the N>1
path of gaudi_amd.dist (shard -> sample -> single gather) with the numpy oracle standing in for the GPU sampler
and the host Philox twin providing the global-index-keyed noise."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from gaudi_amd import dist as gdist  # noqa: E402
from gaudi_amd import synth  # noqa: E402
from gaudi_amd.philox import philox_normal  # noqa: E402
from oracle import gaudi_oracle as O  # noqa: E402  (tests may use the oracle)


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    T, seed = 6, 77
    eargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T)
    pargs = synth.pred_args(nf=36, n_layers=2)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=11)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=12)
    nodes = [5, 7, 3, 7, 6]  # global batch, padded to the global max (7) on every shard
    nm, em = O.build_masks(nodes, max(nodes), False)
    B, N = nm.shape[0], nm.shape[1]
    w = O.target_max_gap_weights(5)

    def sample_fn(nm_s, em_s, offset):
        Bs = nm_s.shape[0]
        noise = philox_normal(seed, offset, Bs, N * 4, 0, T + 2).reshape(T + 2, Bs, N, 4)
        x, h, _ = O.sample(esd, eargs, nm_s[:, :, None], em_s, noise, pred_sd=psd, pcfg=pargs, target_w=w, scale=0.6)
        return x, h

    lo, hi, x, h = gdist.sample_sharded(sample_fn, nm, em, rank, world)
    xs, hs = gdist.gather_to_all(x, h, B, N, 1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=xs, h=hs, lo=lo, hi=hi)
    if rank == 0:
        x_full, h_full = sample_fn(nm.reshape(B, N), em.reshape(B, N, N), 0)
        np.savez(os.path.join(out_dir, "unsharded.npz"), x=x_full, h=h_full)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
