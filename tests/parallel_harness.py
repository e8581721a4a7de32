"""TEST HARNESS (round 1's Python orchestrator; the product's multi-GPU driver is C++: csrc/host/dna_adjust_dist.cpp).

Multi-GPU schedules of the phased adjustment over the per-block C entry points: one process per GPU,
torch.distributed.  Used by tests/test_parallel_gloo.py (numpy backend under gloo), tests/test_gpu_adjust.py and tools/.

1. The condensed schedule (default; settings.schur_carry, dna_adjust_phased.cpp "the condensed schedule").  Per iteration:

  (A) every block is condensed to the stations it shares with its neighbours -- one partial elimination (~n^3/3) per block,
      no dependency between blocks: block k runs on rank owner(k)
  exchange  the condensed systems ((3 * shared stations)^2 + 3 * shared stations doubles, padded to 128 -- 30 MB for a 317-station
      junction row) are broadcast by their owners (async broadcasts, one per block): a real collective, the only data-path one
  (B) the reference's forward and reverse chains, on the condensed blocks: a few thousand unknowns each, run redundantly
      on every rank (identical arithmetic, identical results; cheaper than shipping the junction matrices a second time)
  (C) the one full inverse per block whose result is rigorous (forward for a last block, reverse for a first block,
      combination otherwise), again on rank owner(k), no dependency between blocks
  sync      rigorous coordinates: one all_reduce(sum) of a 3*stations vector; largest correction: one all_gather

  Both large phases shard by block, so B blocks scale to min(B, N) GPUs; the chain (B) is the serial remainder.

2. The reference's schedule (settings.schur_carry = 0): the phased blocks form a chain (forward k -> k+1, reverse k -> k-1, combine(k) needs both), so the path shards the
way the reference's multi-thread mode does (dnaadjust-multi.cpp:92-244): a forward chain, a reverse chain and the
combination solves.  Per iteration:

  phase 1   rank FWD runs the forward chain, rank REV the reverse chain (concurrently, no communication)
  exchange  the junction matrices + junction estimates that the combination solves need travel point-to-point
            (one grouped batch_isend_irecv; each payload is (3|JSL|)^2 + 3|JSL| doubles, padded to 128):
            v_junctionVariancesFwd_/v_junctionEstimatesFwd_[k-1] from FWD, v_junctionVariances_/v_junctionEstimatesRev_[k] from REV
  phase 2   the combination solves of the intermediate blocks, round-robin over ALL ranks
  sync      rigorous coordinates of every block: one all_reduce(sum) of a 3*stations vector; largest correction:
            one all_gather of a scalar per rank (the convergence test of dnaadjust.cpp:2639 on identical data everywhere)

Rigorous variance matrices stay on the rank that produced them.  Critical path per iteration: B + ceil((B-2)/N) solves
instead of 3B-2, i.e. strong scaling saturates near 2.5-2.9x; more GPUs need an intra-block distributed inverse.

The compute is behind a small backend interface so that the schedule and the messaging can be exercised on CPU
(gloo) with a numpy backend (tests/test_parallel_gloo.py); the product backend is DeviceBlockBackend (HIP, through the
dna_adjust facade).  With NCCL (= RCCL on ROCm) the payloads are device tensors and travel over xGMI.
"""
import ctypes as C
import time

import numpy as np


class DeviceBlockBackend:
    """Block steps on this rank's GPU: dna_adjust::Phased* through include/dnaadjust_c.h."""

    def __init__(self, settings, comm_device):
        import torch
        from dynadjust_amd import adjust
        self.torch = torch
        self.adj = adjust.DnaAdjust()
        self.adj.PrepareAdjustment(settings)
        self.lib = self.adj.lib
        self.h = self.adj.h
        self.comm_device = comm_device
        self.n_blocks = self.adj.blockCount()
        self._buf = {}

    def close(self):
        self.adj.close()

    def _chk(self, rc):
        self.adj._chk(rc)

    def flags(self, k):
        f, l, i = C.c_int(), C.c_int(), C.c_int()
        self.lib.dnaadj_block_flags(self.h, k, C.byref(f), C.byref(l), C.byref(i))
        return bool(f.value), bool(l.value), bool(i.value)

    def n_stations(self, k):
        return self.lib.dnaadj_block_station_count(self.h, k)

    def stations(self, k):
        return self.adj.block_stations(k)

    def begin_iteration(self):
        self._chk(self.lib.dnaadj_phased_begin_iteration(self.h))

    def _step(self, fn, k):
        mv = C.c_double()
        self._chk(fn(self.h, k, C.byref(mv)))
        return mv.value

    def forward_block(self, k):
        return self._step(self.lib.dnaadj_phased_forward_block, k)

    def reverse_block(self, k):
        return self._step(self.lib.dnaadj_phased_reverse_block, k)

    def combine_block(self, k):
        return self._step(self.lib.dnaadj_phased_combine_block, k)

    def finalise_block(self, k):
        self._chk(self.lib.dnaadj_phased_finalise_block(self.h, k))

    def note_correction(self, mv):
        self._chk(self.lib.dnaadj_phased_note_correction(self.h, float(mv)))

    def max_correction(self):
        return self.adj.GetMaxCorrection()

    def end_iteration(self):
        it = C.c_int()
        self._chk(self.lib.dnaadj_phased_end_iteration(self.h, C.byref(it)))
        return bool(it.value)

    def finish(self):
        st = C.c_int()
        self._chk(self.lib.dnaadj_phased_finish(self.h, C.byref(st)))
        return st.value

    # ---- condensed schedule ----
    def condensed(self):
        return bool(self.lib.dnaadj_condensed_schedule(self.h))

    def condense_block(self, k):
        self._chk(self.lib.dnaadj_phased_condense_block(self.h, k))

    def condensed_forward(self, k):
        self._chk(self.lib.dnaadj_phased_condensed_forward(self.h, k))

    def condensed_reverse(self, k):
        self._chk(self.lib.dnaadj_phased_condensed_reverse(self.h, k))

    def rigorous_block(self, k):
        return self._step(self.lib.dnaadj_phased_rigorous_block, k)

    # lists of blocks: spread over the two chains of this GPU when settings.multi_thread is on
    def condense_blocks(self, blocks):
        a = np.ascontiguousarray(blocks, dtype=np.uint32)
        self._chk(self.lib.dnaadj_phased_condense_blocks(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size))

    def condensed_chains(self):
        self._chk(self.lib.dnaadj_phased_condensed_chains(self.h))

    def rigorous_blocks(self, blocks):
        a = np.ascontiguousarray(blocks, dtype=np.uint32)
        self._chk(self.lib.dnaadj_phased_rigorous_blocks(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size))

    def condensed_tensor(self, k):
        key = ("c", k)
        if key not in self._buf:
            n = self.lib.dnaadj_condensed_payload_doubles(self.h, k)
            self._buf[key] = self.torch.empty(n, dtype=self.torch.float64, device=self.comm_device) if n else None
        return self._buf[key]

    def export_condensed(self, k):
        t = self.condensed_tensor(k)
        if t is not None:
            self._chk(self.lib.dnaadj_condensed_export(self.h, k, C.c_void_p(t.data_ptr())))
        return t

    def import_condensed(self, k, t):
        self._chk(self.lib.dnaadj_condensed_import(self.h, k, C.c_void_p(t.data_ptr())))

    def junction_tensor(self, kind, k):
        """communication buffer for the junction payload of block k (device tensor with NCCL, host tensor with gloo)"""
        key = (kind, k)
        if key not in self._buf:
            n = self.lib.dnaadj_junction_payload_doubles(self.h, k)
            self._buf[key] = self.torch.empty(n, dtype=self.torch.float64, device=self.comm_device)
        return self._buf[key]

    def export_junction(self, kind, k):
        t = self.junction_tensor(kind, k)
        self._chk(self.lib.dnaadj_junction_export(self.h, kind, k, C.c_void_p(t.data_ptr())))
        return t

    def import_junction(self, kind, k, t):
        self._chk(self.lib.dnaadj_junction_import(self.h, kind, k, C.c_void_p(t.data_ptr())))

    def get_coords(self, k):
        out = np.empty(3 * self.n_stations(k), dtype=np.float64)
        self._chk(self.lib.dnaadj_block_get_coords(self.h, k, 2, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def set_coords(self, k, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        self._chk(self.lib.dnaadj_block_set_coords(self.h, k, xyz.ctypes.data_as(C.POINTER(C.c_double))))


class PhasedSchedule:
    """who does what: static, identical on every rank"""

    def __init__(self, flags, world):
        self.B = len(flags)
        self.flags = flags
        self.world = world
        self.fwd_rank = 0
        self.rev_rank = 1 if world > 1 else 0
        self.intermediate = [k for k, (f, l, i) in enumerate(flags) if not (f or l or i)]
        self.combine_owner = {k: idx % world for idx, k in enumerate(self.intermediate)}

    def final_owner(self, k):
        f, l, i = self.flags[k]
        if l or i:
            return self.fwd_rank      # rigorous from the forward pass (dnaadjust.cpp:3033)
        if f:
            return self.rev_rank      # rigorous from the reverse pass
        return self.combine_owner[k]


def _exchanging(world):
    """collectives run when there is more than one rank -- or when DNAGPU_FORCE_DISTRIBUTED=1 asks for the whole N > 1 path
    (export, collective, import) with however many ranks there are: the way to exercise the RCCL transport on a 1-GPU box"""
    import os
    return world > 1 or bool(int(os.environ.get("DNAGPU_FORCE_DISTRIBUTED", "0")))


def _host_wait(works, dev):
    """Work.wait() on an NCCL work only orders torch's current stream behind the collective; the block steps run on the
    library's own HIP streams, so the host has to see the end of the transfer before the payload is imported."""
    for w in works:
        w.wait()
    if getattr(dev, "type", "cpu") == "cuda":
        import torch
        torch.cuda.current_stream(dev).synchronize()


def block_owners(costs, world):
    """static, identical on every rank: longest-processing-time-first over the per-block cost (n^3)"""
    load = [0.0] * world
    owner = [0] * len(costs)
    for k in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k] = r
        load[r] += costs[k]
    return owner


def _sync_coordinates(backend, dist, rank, world, owner_of, offs, dev):
    """rigorous coordinates of every block and the largest correction, on every rank"""
    import torch
    B = backend.n_blocks
    flat = np.zeros(int(offs[B]), dtype=np.float64)
    for k in range(B):
        if owner_of(k) == rank:
            flat[offs[k]:offs[k + 1]] = backend.get_coords(k)
    t = torch.from_numpy(flat).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.cpu().numpy()
    for k in range(B):
        if owner_of(k) != rank:
            backend.set_coords(k, flat[offs[k]:offs[k + 1]])
    mine = torch.tensor([backend.max_correction()], dtype=torch.float64, device=dev)
    allc = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(allc, mine)
    for c in allc:
        backend.note_correction(float(c.item()))


def distributed_statistics(backend, dist, rank, world, owner_of):
    """dna_adjust::GenerateStatistics when the rigorous variances are spread over the ranks: every rank computes the precisions
    of the adjusted measurements, the per-record statistics and the chi-square terms of the blocks it owns; one all_reduce(sum)
    of (chi-square, outliers) and one of the per-record arrays give every rank the whole picture; the global figures follow."""
    import torch
    adj = backend.adj
    lib, h = backend.lib, backend.h
    backend._chk(lib.dnaadj_statistics_prepare(h))
    mine = np.ascontiguousarray([k for k in range(backend.n_blocks) if owner_of(k) == rank], dtype=np.uint32)
    backend._chk(lib.dnaadj_statistics_blocks(h, mine.ctypes.data_as(C.POINTER(C.c_uint32)), mine.size))
    if _exchanging(world):
        chi, out = C.c_double(), C.c_uint32()
        lib.dnaadj_statistics_get_partial(h, C.byref(chi), C.byref(out))
        n = adj.lib.dnaadj_measurement_record_count(h)
        rec = np.zeros((n, 9), dtype=np.float64)
        backend._chk(lib.dnaadj_record_statistics_get(h, rec.ctypes.data_as(C.POINTER(C.c_double)), n))
        dev = backend.comm_device
        t = torch.from_numpy(np.concatenate([[chi.value, float(out.value)], rec.ravel()])).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        a = t.cpu().numpy()
        backend._chk(lib.dnaadj_statistics_set_partial(h, float(a[0]), int(round(a[1]))))
        rec = np.ascontiguousarray(a[2:].reshape(n, 9))
        backend._chk(lib.dnaadj_record_statistics_set(h, rec.ctypes.data_as(C.POINTER(C.c_double)), n))
    backend._chk(lib.dnaadj_statistics_finish(h))


def run_phased_condensed(backend, dist, rank, world, max_iterations=10):
    """the condensed schedule across `world` ranks; returns (status, iterations, per-iteration corrections, owners)"""
    B = backend.n_blocks
    owner = block_owners([float(backend.n_stations(k)) ** 3 for k in range(B)], world)
    offs = np.zeros(B + 1, dtype=np.int64)
    for k in range(B):
        offs[k + 1] = offs[k] + 3 * backend.n_stations(k)
    corrections = []
    import os
    timing = bool(int(os.environ.get("DNAGPU_DIST_TIMING", "0")))     # diagnostic: where an iteration's wall time goes, per rank
    for _ in range(max_iterations):
        tm = [time.perf_counter()]
        backend.begin_iteration()
        mine = [k for k in range(B) if owner[k] == rank]
        # (A) + exchange: one broadcast per block from its owner, all in flight together
        backend.condense_blocks(mine)
        tm.append(time.perf_counter())
        if _exchanging(world):
            pending = []
            for k in range(B):
                t = backend.export_condensed(k) if owner[k] == rank else backend.condensed_tensor(k)
                if t is not None:
                    pending.append((k, t, dist.broadcast(t, src=owner[k], async_op=True)))
            _host_wait([w for _, _, w in pending], backend.comm_device)
            for k, t, _ in pending:
                if owner[k] != rank or world == 1:       # (world == 1: forced exchange, re-importing its own payload)
                    backend.import_condensed(k, t)
        tm.append(time.perf_counter())
        # (B) the two chains on the condensed blocks, everywhere
        backend.condensed_chains()
        tm.append(time.perf_counter())
        # (C)
        backend.rigorous_blocks(mine)
        tm.append(time.perf_counter())
        if _exchanging(world):
            _sync_coordinates(backend, dist, rank, world, lambda k: owner[k], offs, backend.comm_device)
        corrections.append(backend.max_correction())
        tm.append(time.perf_counter())
        more = backend.end_iteration()
        tm.append(time.perf_counter())
        if timing:
            names = ("condense", "exchange", "chains", "rigorous", "coordinates", "end_iteration")
            print("rank %d iteration %d: " % (rank, len(corrections)) +
                  ", ".join("%s %.1f ms" % (n, 1e3 * (tm[i + 1] - tm[i])) for i, n in enumerate(names)), flush=True)
        if not more:
            break
    status = backend.finish()
    return status, len(corrections), corrections, owner


def run_phased(backend, dist, rank, world, max_iterations=10):
    """AdjustPhased (dnaadjust.cpp:2579) across `world` ranks; returns (status, iterations, per-iteration corrections)."""
    import torch
    if getattr(backend, "condensed", lambda: False)():
        return run_phased_condensed(backend, dist, rank, world, max_iterations)[:3]
    flags = [backend.flags(k) for k in range(backend.n_blocks)]
    sch = PhasedSchedule(flags, world)
    B = sch.B
    offs = np.zeros(B + 1, dtype=np.int64)
    for k in range(B):
        offs[k + 1] = offs[k] + 3 * backend.n_stations(k)
    corrections = []
    dev = backend.comm_device
    for _ in range(max_iterations):
        backend.begin_iteration()
        # ---- phase 1: the two chains ------------------------------------------------------------------
        if rank == sch.fwd_rank:
            for k in range(B):
                backend.forward_block(k)          # notes the correction of the last / isolated block itself
        if rank == sch.rev_rank:
            for k in range(B - 1, -1, -1):
                f, l, i = flags[k]
                if i:
                    continue
                mv = backend.reverse_block(k)
                if f and not l:                   # first block of a network: rigorous now
                    backend.note_correction(mv)
                    backend.finalise_block(k)
        # ---- exchange the junction payloads of the combination solves ----------------------------------
        if _exchanging(world):
            ops, recvs = [], []
            for k in sch.intermediate:
                o = sch.combine_owner[k]
                for kind, src, blk in ((0, sch.fwd_rank, k - 1), (1, sch.rev_rank, k)):
                    if src == o:
                        continue
                    if rank == src:
                        ops.append(dist.P2POp(dist.isend, backend.export_junction(kind, blk), o))
                    elif rank == o:
                        t = backend.junction_tensor(kind, blk)
                        ops.append(dist.P2POp(dist.irecv, t, src))
                        recvs.append((kind, blk, t))
            if ops:
                _host_wait(dist.batch_isend_irecv(ops), dev)
            for kind, blk, t in recvs:
                backend.import_junction(kind, blk, t)
        # ---- phase 2: combination solves -----------------------------------------------------------------
        for k in sch.intermediate:
            if sch.combine_owner[k] == rank:
                mv = backend.combine_block(k)
                backend.note_correction(mv)
                backend.finalise_block(k)
        # ---- rigorous coordinates and the largest correction, on every rank -------------------------------
        if _exchanging(world):
            _sync_coordinates(backend, dist, rank, world, sch.final_owner, offs, dev)
        corrections.append(backend.max_correction())
        if not backend.end_iteration():
            break
    status = backend.finish()
    return status, len(corrections), corrections


def bench_distributed(folder, name, phased, args, dist, rank, world, local_rank, dist_backend="nccl"):
    """bench.py --gpus N (N > 1): the same network on N ranks, strong scaling; rank 0 returns the JSON dict."""
    import torch
    from dynadjust_amd import adjust
    if not phased:
        raise SystemExit("the simultaneous adjustment does not shard: run it with --gpus 1")
    dev = torch.device("cuda", local_rank) if dist_backend == "nccl" else torch.device("cpu")   # where the payloads live
    import os
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, device=local_rank,
                               multi_thread=bool(int(os.environ.get("DNAGPU_MULTI_THREAD", "1"))),
                               schur_carry=not getattr(args, "reference_schedule", False),
                               keep_factors=not getattr(args, "no_keep_factors", False))
    be = DeviceBlockBackend(p, dev)
    a = be.adj
    lib, ctx = a.lib, a.device_context()
    condensed = be.condensed()

    def one_step():
        a.ResetAdjustment()
        st, its, corr = run_phased(be, dist, rank, world, max_iterations=p.max_iterations)
        if st != adjust.ADJUST_SUCCESS:
            raise SystemExit(f"adjustment did not converge (status {st})")
        return its

    for _ in range(args.warmup):
        one_step()
    lib.dnagpu_profile_enable(ctx, 1)
    lib.dnagpu_profile_reset(ctx)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        its = one_step()
    lib.dnagpu_sync(ctx)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    prof_f, prof_ms, prof_n = C.c_double(), C.c_double(), C.c_uint64()
    lib.dnagpu_profile_get(ctx, C.byref(prof_f), C.byref(prof_ms), C.byref(prof_n))
    lib.dnagpu_profile_enable(ctx, 0)
    # per rank: algorithmic flops of its own steps (the chains on the condensed blocks run on every rank: their < 0.5 % is
    # counted on every rank), HIP-event time of its GEMM launches; the reference-equivalent count (n^3 per Solve() of the
    # reference's schedule) is the same on every rank with the condensed schedule and split over the ranks otherwise
    mine = torch.tensor([a.algorithmic_flops(), prof_ms.value, float(a.solve_count()), a.solve_flops()], dtype=torch.float64, device=dev)
    allv = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(allv, mine)
    allv = [v.cpu().numpy() for v in allv]
    dt = float(dt.item())
    stations = a.lib.dnaadj_station_count(a.h)
    out = None
    if rank == 0:
        alg = sum(float(v[0]) for v in allv)
        # the reference's own work for this adjustment: a Solve() (n^3) per block in the forward pass, one in the reverse pass
        # unless the block is isolated, one combination solve for every intermediate block -- per iteration
        solves, ref = 0, 0.0
        for k in range(be.n_blocks):
            f, l, i = be.flags(k)
            m = 1 + (0 if i else 1) + (0 if (f or l or i) else 1)
            solves += its * m
            ref += its * m * (3.0 * be.n_stations(k)) ** 3
        busiest = max(range(world), key=lambda r: allv[r][1])
        gemm_ms = float(allv[busiest][1])
        # the flop counters restart with every step (ResetAdjustment), the event time accumulates over the timed steps
        achieved = (float(allv[busiest][0]) / 1e12) / (gemm_ms / args.steps / 1e3) if gemm_ms > 0 else 0.0
        par = (f"condensed schedule: blocks condensed and solved rigorously on their owner rank ({be.n_blocks} blocks over {world} ranks, "
               f"{os.environ.get('DNAGPU_CHAINS', '4') + ' chains' if p.multi_thread else 'one chain'} per GPU), condensed systems broadcast over RCCL, chains on the condensed "
               "blocks on every rank") if condensed else (
               f"forward chain on rank 0, reverse chain on rank 1, combination solves round-robin over {world} ranks; "
               "junction matrices point-to-point over RCCL")
        out = {
            "metric": "stations adjusted/sec + Cholesky TFLOP/s, phased adjustment, 1/2/4/8 MI355X",
            "value": stations * args.steps / dt,
            "unit": "stations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"stations": stations, "blocks": be.n_blocks, "iterations_to_converge": its, "mode": "phased",
                       "solves_per_step": solves, "schur_carry": condensed,
                       "keep_factors": bool(a.completion_count()), "parallelism": par},
            "cholesky_tflops": (alg / 1e12) / (dt / args.steps),
            "reference_equivalent_tflops": (ref / 1e12) / (dt / args.steps),
            "roofline": {
                "kernel": "gemm_f64_dma_kernel (v_mfma_f64_16x16x4_f64 tile GEMM behind potrf/trtri/lauum)", "bound": "mfma",
                "achieved": achieved, "peak": 78.6, "unit": "TFLOP/s", "frac": achieved / 78.6, "traffic": None,
                "rank": busiest, "gemm_ms_per_step": gemm_ms / args.steps,
                "note": "busiest rank: algorithmic flops of its steps / HIP-event time of its GEMM launches",
            },
        }
    # after the timed region: statistics across the ranks and the distance from the truth the generator kept
    try:
        if condensed:
            owner = block_owners([float(be.n_stations(k)) ** 3 for k in range(be.n_blocks)], world)
            final_owner = lambda k: owner[k]
        else:
            final_owner = PhasedSchedule([be.flags(k) for k in range(be.n_blocks)], world).final_owner
        distributed_statistics(be, dist, rank, world, final_owner)
        if out is not None:
            truth = np.fromfile(os.path.join(folder, name + ".truth"), dtype=np.float64).reshape(-1, 3)
            xyz = a.adjusted_coordinates(stations)
            out["check"] = {"sigma_zero": a.GetSigmaZero(), "degrees_of_freedom": a.GetDegreesOfFreedom(),
                            "max_abs_error_vs_truth_m": float(np.abs(xyz - truth).max()), "global_test": int(a.GetTestResult()),
                            "chi_squared_limits": [a.GetChiSquaredLowerLimit(), a.GetChiSquaredUpperLimit()]}
    except Exception as e:                       # diagnostic only
        if out is not None:
            out["check"] = {"error": str(e)}
    be.close()
    return out
