"""One rank of a process-per-GPU adjustment (tests/test_gpu_multi.py::test_process_per_gpu_bootstrap, tests/test_gpu_distributed.py::
test_processes_sharing_the_gpu*): RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment like under torchrun or mpirun,
nothing else shared between the processes.  PrepareAdjustment() makes the communicator itself -- RCCL: rank 0's ncclUniqueId reaches the
others over TCP (dist_comm.cpp tcp_share_unique_id); DNAGPU_DIST_TRANSPORT=shared: processes on ONE GPU, host-staged over TCP
(dist_comm_shared.cpp) -- and rank 0 leaves the results in <folder>/result.npz.
WORKER_DEVICE: the HIP device of this rank (default: its rank).  WORKER_EXPECT: the transport dist_info() must report (default rccl).
WORKER_SETTINGS: JSON of extra ProjectSettings arguments.  WORKER_DIE_AFTER_S (with WORKER_DIE_RANK): that rank kills itself that long after
PrepareAdjustment, in the middle of the adjustment -- the others must come back with an exception, not hang."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from dynadjust_amd import adjust  # noqa: E402


def main():
    folder, name = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ.get("WORKER_DEVICE", rank))
    expect = os.environ.get("WORKER_EXPECT", "rccl")
    extra = json.loads(os.environ.get("WORKER_SETTINGS", "{}"))
    a = adjust.DnaAdjust()
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, device=device, dist_rank=rank, dist_world=world,
                               output_folder=os.path.join(folder, "out"), **extra)
    a.PrepareAdjustment(p)
    r, w, transport = a.dist_info()
    assert (r, w, transport) == (rank, world, expect), (r, w, transport)
    if "WORKER_DIE_AFTER_S" in os.environ and rank == int(os.environ.get("WORKER_DIE_RANK", "1")):
        threading.Thread(target=lambda: (time.sleep(float(os.environ["WORKER_DIE_AFTER_S"])), os._exit(9)), daemon=True).start()
    t0 = time.perf_counter()
    try:
        st = a.AdjustNetwork()
    except adjust.NetAdjustException as e:
        print(f"rank {rank}: exception after {time.perf_counter() - t0:.1f} s: {e}", flush=True)
        sys.exit(3)
    a.GenerateStatistics()
    a.SerialiseAdjustedVarianceMatrices()          # collective: the other ranks' variance matrices travel to rank 0
    if rank == 0:
        B = a.blockCount()
        plan = a.memory_plan()
        out = {"status": st, "iterations": a.CurrentIteration(), "chi": a.GetChiSquared(), "rccl_ranks": a.device_instance_stats(0)["rccl_ranks"],
               "owners": np.array([a.block_owner(k) for k in range(B)]), "exchanged_bytes": a.exchange_stats()["bytes"],
               "factors_parked": plan["blocks_packing_their_factor"], "factors_taken": plan["factors_taken_from_their_packed_copy"],
               "factors_made_again": plan["factors_made_again"], "staged_host_bytes": plan["staged_variances_host_bytes"]}
        for k in range(B):
            out[f"est_{k}"] = a.block_estimates(k)
        np.savez(os.path.join(folder, "result.npz"), **out)
    a.close()


if __name__ == "__main__":
    main()
