"""One rank of a process-per-GPU adjustment (tests/test_gpu_multi.py::test_process_per_gpu_bootstrap): RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT from the environment like under torchrun or mpirun, nothing else shared between the processes.
PrepareAdjustment() makes the RCCL communicator itself -- rank 0's ncclUniqueId reaches the others over TCP (dist_comm.cpp
tcp_share_unique_id) -- and rank 0 leaves the results in <folder>/result.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from dynadjust_amd import adjust  # noqa: E402


def main():
    folder, name = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    a = adjust.DnaAdjust()
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, device=rank, dist_rank=rank, dist_world=world,
                               output_folder=os.path.join(folder, "out"))
    a.PrepareAdjustment(p)
    r, w, transport = a.dist_info()
    assert (r, w, transport) == (rank, world, "rccl"), (r, w, transport)
    st = a.AdjustNetwork()
    a.GenerateStatistics()
    a.SerialiseAdjustedVarianceMatrices()          # collective: the other ranks' variance matrices travel to rank 0
    if rank == 0:
        B = a.blockCount()
        out = {"status": st, "iterations": a.CurrentIteration(), "chi": a.GetChiSquared(), "rccl_ranks": a.device_instance_stats(0)["rccl_ranks"],
               "owners": np.array([a.block_owner(k) for k in range(B)])}
        for k in range(B):
            out[f"est_{k}"] = a.block_estimates(k)
        np.savez(os.path.join(folder, "result.npz"), **out)
    a.close()


if __name__ == "__main__":
    main()
