"""Worker for tests/test_gpu_round3.py::test_rccl_gather_world1: ONE rank, backend "nccl" (= RCCL on ROCm) on cuda:0.
Runs the real Engine on a small guided batch, then the exact distributed code of bench.py --gpus N / gaudi_amd.dist:
init_process_group("nccl", device_id=...), a device-tensor all_gather (gather_to_all), the plan check and the MAX
all_reduce bench.py times with.  A 1-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so this
is the most the hardware at hand can execute of the N > 1 path; the 2-rank logic is covered under gloo."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gaudi_amd import dist as gdist  # noqa: E402
from gaudi_amd import synth  # noqa: E402
from gaudi_amd.engine import Engine  # noqa: E402
from gaudi_amd.sampling_edm import build_masks  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    T = 20
    eargs = synth.edm_args(nf=64, n_layers=3, diffusion_steps=T)
    pargs = synth.pred_args(nf=60, n_layers=3)
    eng = Engine(0)
    eng.load_edm(eargs, synth.synth_edm_state_dict(eargs, 1, seed=11))
    eng.load_predictor(pargs, synth.synth_predictor_state_dict(pargs, 1, 5, seed=12))
    nodes = np.array([5, 7, 3, 11, 6, 11, 2])
    nm3, em_flat, _ = build_masks(nodes, int(nodes.max()), False)
    B, N = nm3.shape[0], nm3.shape[1]
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    w = np.array([0, -1, 0, 0, 0], np.float32)

    def sample_fn(nm_s, em_s, offset):
        x, h, _ = eng.sample(nm_s, em_s, seed=5, sample_offset=offset, target_w=w, scale=0.6)
        return x, h

    lo, hi, x, h = gdist.sample_sharded(sample_fn, nm, em, 0, 1, engine=eng)
    xs, hs = gdist.gather_to_all(x, h, B, N, 1, device=dev)  # RCCL all_gather of a device tensor
    plan = gdist.check_same_plan(eng, device=dev)
    tt = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
    dist.barrier()
    torch.cuda.synchronize()
    info = dict(lo=lo, hi=hi, same_x=bool(np.array_equal(xs, x)), same_h=bool(np.array_equal(hs, h)),
                finite=bool(np.isfinite(xs).all()), plan=list(plan), allreduce=float(tt.item()),
                backend=dist.get_backend(), rccl_version=list(torch.cuda.nccl.version()),
                hip=torch.version.hip, device=torch.cuda.get_device_name(0))
    with open(os.path.join(out_dir, "nccl_world1.json"), "w") as f:
        json.dump(info, f)
    print("RCCL world-1 run:", json.dumps(info), flush=True)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        with open(os.path.join(sys.argv[1], "err0.txt"), "w") as f:
            f.write(traceback.format_exc())
        raise
