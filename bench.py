#!/usr/bin/env python
"""bench.py -- stations adjusted / s of the phased least-squares adjustment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4|cfg4_slice|cfg3_ragged|smallblocks|dnasegment150|dnasegment150_10x|small]

A "step" is one complete dna_adjust::AdjustNetwork() (phased: forward + reverse + combine sweeps,
iterated to the reference's convergence threshold) on one synthetic network whose matrices and
measurements are already resident in HBM when the timed region starts (dnaadj_reset puts the
coordinates back between steps; file loading and PrepareAdjustment are outside the timed region).

Default workload (N = 1): BASELINE.json configs[2] "synthetic 100k-station / 800k-measurement
network, phased adjustment, 16 blocks, 1 x MI355X" -- the largest named phased configuration that
fits one GPU (the metric is quoted on the phased adjustment).

Other workloads: the reference's default cut (dnasegment150: 666 blocks of 150 stations; dnasegment150_10x: a project of ten such networks,
1M stations), mixed small blocks (smallblocks), cfg2 / cfg4 / cfg4_slice of BASELINE.json, uneven strips (cfg3_ragged).

One JSON line is printed by rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.
# Four chain streams fill them: with RCCL's stream (N > 1) or the staged mode's copy streams beside them, the forward and the
# reverse chain of the condensed schedule landed on one queue (chain phase 27 -> 48 ms per iteration).  Must be set before the
# HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# RCCL between processes shares device memory by dmabuf handles; the pool's host driver supports no other kind (without this,
# ncclCommInitRank fails with "hipIpcGetMemHandle: invalid argument").  Set on the GPU boxes already; kept here for a bare environment.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOADS = {
    # name: (rows, cols, baselines, blocks, phased, description)
    "cfg3": (316, 317, 266666, 16, True, "synthetic 100k-station / 800k-measurement network, phased adjustment, 16 blocks"),
    "cfg2": (100, 100, 26666, 1, False, "synthetic 10k-station / 80k-measurement network, simultaneous adjustment"),
    # BASELINE.json configs[3]: needs the rigorous variances of 128 blocks of n ~ 27 000 (0.75 TB): 4 or more GPUs
    "cfg4": (1000, 1000, 2666666, 128, True, "synthetic 1M-station / 8M-measurement network, phased adjustment, 128 blocks"),
    # 16 of cfg4's 128 strips: the same block size (n ~ 27 000, 1000-station junction rows) on one GPU
    "cfg4_slice": (125, 1000, 333333, 16, True, "synthetic 125k-station / 1M-measurement network, phased adjustment, 16 blocks of cfg4's size"),
    "small": (60, 60, 9600, 4, True, "synthetic 3.6k-station / 28.8k-measurement network, phased adjustment, 4 blocks (smoke size)"),
    # uneven segmentations (what dnasegment makes of a real network: no two blocks alike, dnasegment.cpp:235-348):
    # cfg3 with strip heights drawn +-30 % around the mean (blocks of n ~ 14 000 ... 26 000)
    "cfg3_ragged": (316, 317, 266666, 16, True, "synthetic 100k-station / 800k-measurement network, phased adjustment, 16 blocks of uneven size (strip heights +-30 %)"),
    # the same station count cut the way dnasegment's defaults would (150 stations per block and up, dnaoptions.hpp:381-382): strips of 1 ... 10
    # rows of 150 stations -> ~120 blocks of 150 ... 1 500 inner + 150 junction stations (n = 900 ... 4 950)
    "smallblocks": (668, 150, 266666, 0, True, "synthetic 100k-station / 800k-measurement network, phased adjustment, ~120 blocks of 150 ... 1 500 inner stations (dnasegment-like cut)"),
    # the reference's DEFAULT operating point: dnasegment cuts blocks of 150 stations unless told otherwise (min_inner_stations(150),
    # max_total_stations(150): include/config/dnaoptions.hpp:382, dnasegment.cpp:594-596).  100 000 stations on a grid 50 wide, strips of
    # 3 rows: 667 blocks of 150 inner + 50 junction stations, n = 600 unknowns each -- hundreds of small dense systems, not a few large ones
    "dnasegment150": (2000, 50, 266666, 0, True, "synthetic 100k-station / 800k-measurement network, phased adjustment, 667 blocks of 150 inner + 50 junction stations (dnasegment's default block size)"),
}
# a PROJECT of the national size at the reference's default cut: ten contiguous networks (network ids) of the dnasegment150 kind in one set of files,
# 1M stations in 6 660 blocks.  (One strip of 20 000 x 50 stations would be a different problem: its normal equations are so ill-conditioned along
# the strip that the iterations act as iterative refinement and do not reach the 0.5 mm threshold in ten.)
WORKLOADS["dnasegment150_10x"] = (2000, 50, 266666, 0, True, "synthetic 1M-station / 8M-measurement project, phased adjustment, ten networks of 666 blocks of 150 inner + 50 junction stations each (dnasegment's default block size)")
# extra arguments of the generator per workload (dnasynth_spec: ragged, rows_lo, rows_hi)
WORKLOAD_KW = {"cfg3_ragged": {"ragged": 0.3}, "smallblocks": {"rows_lo": 1, "rows_hi": 10}, "dnasegment150": {"rows_lo": 3, "rows_hi": 3},
               "dnasegment150_10x": {"rows_lo": 3, "rows_hi": 3}}
WORKLOAD_COPIES = {"dnasegment150_10x": 10}
FP64_MFMA_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64: 2048 flop / 64 clk)


def traffic_from_profile(workload):
    """HBM-side bytes per launch of the dominant kernel.  PMC counters cannot be read from inside the process, so this is
    the figure of the committed `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command
    (profiles/r0N_hbm_traffic.json, made by tools/pmc_traffic_json.py) -- quoted only while the device sources still are the ones
    the passes were made with (their hash is in the record): (bytes, note)"""
    import glob
    from tools.pmc_traffic_json import csrc_sha16
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):     # newest round first
        try:
            rec = json.load(open(path))
            if rec.get("workload") != workload:
                continue
            if rec.get("csrc_sha16") != csrc_sha16():
                return None, f"{os.path.basename(path)} predates the last change to dynadjust_amd/csrc (tools/refresh_profiles.sh makes a new one): not quoted"
            return rec["hbm_bytes_per_launch"], f"{os.path.basename(path)} (device sources {rec['csrc_sha16']})"
        except (OSError, ValueError, KeyError):
            continue
    return None, "no PMC pass of this workload under profiles/"


def _full_run_record(workload):
    """profiles/r0N_oracle_<workload>_run.json: the CPU restatement over the whole workload, once, on a GPU box's host (newest round first)"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_oracle_{workload}_run.json")), reverse=True):
        try:
            rec = json.load(open(path))
            return {"file": os.path.relpath(path, ROOT), "seconds": rec.get("oracle_seconds"), "tflops_reference_equivalent": rec.get("oracle_tflops"),
                    "threads": rec.get("oracle_threads"), "lapack": rec.get("lapack"), "solves": rec.get("oracle_solves")}
        except (OSError, ValueError):
            continue
    return None


def cpu_baseline(workload, iterations, solves_per_step, sum_n3_per_step, stations):
    """Runs _cpu_baseline_sample in a clean subprocess: the MKL runtime must not share a process with torch's
    OpenMP runtime (mixing libiomp5 and libgomp silently corrupts dpotrf results), and MKL_THREADING_LAYER=GNU
    has to be in the environment before libmkl_rt is loaded."""
    import subprocess
    env = dict(os.environ, MKL_THREADING_LAYER="GNU")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload,
           "--cpu-args", json.dumps([iterations, solves_per_step, sum_n3_per_step, stations])]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode == 0 and line:
            return json.loads(line[-1])
        return {"value": None, "unit": "stations/s", "cores": 0, "kind": "port", "sample": "failed: " + (out.stderr or out.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "stations/s", "cores": 0, "kind": "port", "sample": "timed out after 600 s"}


def _cpu_baseline_role(role, folder, lapack_path, threads, cores):
    """One of the two passes of the reference's --multi-thread schedule (dnaadjust-multi.cpp:92-244) as a PROCESS of its own: pinned to
    `cores`, its LAPACK on `threads` threads.  "fwd" times the forward pass; "rev" runs a forward pass first (untimed: the reverse +
    combination pass starts from the forward results) and times the reverse + combination pass.  Both wait for <folder>/go so that the
    timed passes run side by side.  Prints one JSON line."""
    from tests import oracle
    if cores:
        try:
            os.sched_setaffinity(0, set(cores))
        except OSError:
            pass
    lib = oracle.load()
    oracle.use_lapack(lapack_path)
    lib.orc_set_threads(threads)
    net = oracle.Network(os.path.join(folder, "cpu"), True)
    o = oracle.Adjustment(net, True, threads=0)
    o.prepare()
    s0 = n0 = 0
    if role == "rev":
        if lib.orc_adjust_forward_pass(o.h):
            raise RuntimeError(lib.orc_adjust_error(o.h).decode())
        s0, n0 = o.solve_stats()
    open(os.path.join(folder, "ready." + role), "w").close()
    t_wait = time.perf_counter()
    while not os.path.exists(os.path.join(folder, "go")):
        time.sleep(0.005)
        if time.perf_counter() - t_wait > 600:
            raise RuntimeError("the other pass never became ready")
    t0 = time.perf_counter()
    rc = lib.orc_adjust_forward_pass(o.h) if role == "fwd" else lib.orc_adjust_reverse_pass(o.h)
    dt = time.perf_counter() - t0
    if rc:
        raise RuntimeError(lib.orc_adjust_error(o.h).decode())
    s1, n1 = o.solve_stats()
    print(json.dumps({"role": role, "seconds": dt, "solves": int(s1 - s0), "n3": n1 - n0, "t_start": t0, "t_end": t0 + dt}), flush=True)
    o.close()


def _cpu_solve_role(folder, lapack_path, threads, cores, n, idx):
    """One worker of the combine pool (dnaadjust-multi.cpp:644-710): a Solve()-sized dpotrf + dpotri of order n on `threads` LAPACK threads,
    pinned to `cores`, started together with the others.  Prints one JSON line."""
    import numpy as np
    from tests import oracle
    if cores:
        try:
            os.sched_setaffinity(0, set(cores))
        except OSError:
            pass
    lib = oracle.load()
    oracle.use_lapack(lapack_path)
    lib.orc_set_threads(threads)
    rng = np.random.default_rng(idx)
    A = rng.standard_normal((n, 32))
    M = np.asfortranarray(A @ A.T)
    M[np.diag_indices(n)] += float(n)
    del A
    open(os.path.join(folder, f"ready.solve{idx}"), "w").close()
    t_wait = time.perf_counter()
    while not os.path.exists(os.path.join(folder, "go.solve")):
        time.sleep(0.005)
        if time.perf_counter() - t_wait > 600:
            raise RuntimeError("the pool never started")
    t0 = time.perf_counter()
    lib.orc_potrf_lower(n, M.ctypes.data_as(oracle.f64p), n)
    lib.orc_potri_lower(n, M.ctypes.data_as(oracle.f64p), n)
    dt = time.perf_counter() - t0
    print(json.dumps({"role": "solve", "seconds": dt, "t_start": t0, "t_end": t0 + dt}), flush=True)


def _cpu_combine_pool(folder, lapack_path, quota_threads, n, pool):
    """The reference's combine pool as a throughput bound: `pool` Solve()-sized inversions (order n) side by side, quota_threads / pool
    LAPACK threads each, as pinned processes -- what K concurrent combination solves can deliver on this host at best (an upper bound
    on any schedule's reference-equivalent rate, the forward and reverse chains' dependencies ignored).  Returns TFLOP/s (sum n^3 / wall)."""
    import subprocess
    t = max(1, quota_threads // pool)
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    for f in os.listdir(folder):
        if f.startswith("ready.solve") or f == "go.solve":
            os.remove(os.path.join(folder, f))
    env = dict(os.environ, MKL_THREADING_LAYER="GNU", OPENBLAS_NUM_THREADS=str(t), OMP_NUM_THREADS=str(t))
    procs = []
    for i in range(pool):
        cores = avail[i * t:(i + 1) * t] if len(avail) >= pool * t else None
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-role", "solve", "--cpu-args", json.dumps([folder, lapack_path, t, cores, n, i])]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.perf_counter()
    while not all(os.path.exists(os.path.join(folder, f"ready.solve{i}")) for i in range(pool)):
        if any(p.poll() is not None for p in procs) or time.perf_counter() - t0 > 600:
            for p in procs:
                p.kill()
            raise RuntimeError("a pool worker did not start: " + " | ".join((p.stderr.read() or "")[-200:] for p in procs if p.poll() is not None))
        time.sleep(0.01)
    open(os.path.join(folder, "go.solve"), "w").close()
    recs = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        line = [l for l in out.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            raise RuntimeError("a pool worker failed: " + (err or out)[-300:])
        recs.append(json.loads(line[-1]))
    wall = max(r["t_end"] for r in recs) - min(r["t_start"] for r in recs)
    return pool * float(n) ** 3 / wall / 1e12


def _cpu_limits():
    """what the host gives this process: visible cores, the affinity mask, the container's CPU quota (cgroup v2 cpu.max / v1 cfs)"""
    out = {"cores_visible": os.cpu_count(), "affinity": None, "cgroup_cpu_max": None, "quota_cores": None}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        out["cgroup_cpu_max"] = " ".join(txt)
        if txt[0] != "max":
            out["quota_cores"] = float(txt[0]) / float(txt[1])
    except (OSError, IndexError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_cpu_max"] = f"{int(q)} {int(p)}"
            if q > 0:
                out["quota_cores"] = q / p
        except (OSError, ValueError):
            pass
    return out


def _cpu_multi_thread_by_processes(folder, lapack_path, threads):
    """forward pass || reverse + combination pass as two pinned processes with half of the tuned LAPACK threads each (a LAPACK without
    per-thread control -- OpenBLAS -- cannot be split inside one process).  Returns (wall seconds, sum n^3, Solve() calls, note)."""
    import subprocess
    half = max(1, threads // 2)
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    sets = [avail[:half], avail[half:2 * half]] if len(avail) >= 2 * half else [None, None]
    for f in ("go", "ready.fwd", "ready.rev"):
        try:
            os.remove(os.path.join(folder, f))
        except OSError:
            pass
    env = dict(os.environ, MKL_THREADING_LAYER="GNU", OPENBLAS_NUM_THREADS=str(half), OMP_NUM_THREADS=str(half))
    procs = []
    for role, cores in zip(("fwd", "rev"), sets):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-role", role, "--cpu-args", json.dumps([folder, lapack_path, half, cores])]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.perf_counter()
    while not (os.path.exists(os.path.join(folder, "ready.fwd")) and os.path.exists(os.path.join(folder, "ready.rev"))):
        if any(p.poll() is not None for p in procs) or time.perf_counter() - t0 > 600:
            for p in procs:
                p.kill()
            raise RuntimeError("a pass of the multi-thread schedule did not start: " + " | ".join((p.stderr.read() or "")[-200:] for p in procs))
        time.sleep(0.01)
    open(os.path.join(folder, "go"), "w").close()
    recs = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        line = [l for l in out.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            raise RuntimeError("a pass of the multi-thread schedule failed: " + (err or out)[-300:])
        recs.append(json.loads(line[-1]))
    wall = max(r["t_end"] for r in recs) - min(r["t_start"] for r in recs)      # (both clocks are this host's monotonic clock)
    note = (f"forward pass || reverse + combination pass as two processes pinned to disjoint sets of {half} cores, {half} LAPACK threads each "
            f"(forward {recs[0]['seconds']:.1f} s, reverse + combination {recs[1]['seconds']:.1f} s)")
    return wall, recs[0]["n3"] + recs[1]["n3"], recs[0]["solves"] + recs[1]["solves"], note


def _cpu_baseline_sample(workload, iterations, solves_per_step, sum_n3_per_step, stations):
    """The CPU restatement (oracle/) timed on this host's cores in the reference's own parallel schedule.

    1. LAPACK: the MKL runtime (what the reference links) and the OpenBLAS inside the scipy wheel are probed with dpotrf + dpotri at
       n = 8 192 over thread counts up to every visible core; the faster library and its best thread count are used.
    2. Sample: a strip of the workload's grid at the workload's REAL block size (cfg3: 3 blocks of n ~ 20 000; cfg2: the whole
       n = 30 000 block) -- bounded by the block count, not by shrinking n.
    3. Schedule (phased): the reference's --multi-thread mode (dnaadjust-multi.cpp:92-244): forward pass on one thread, reverse +
       combination pass on a second, each calling LAPACK with half of the tuned threads (MKL_Set_Num_Threads_Local); timed as
       two adjustments of the same strip side by side, the second already past its forward pass.  Without per-thread control
       (OpenBLAS) or when it is slower, the sequential schedule on all tuned threads.
    Extrapolated to the workload linearly in sum n^3 of its Solve() calls."""
    import threading
    import numpy as np
    from dynadjust_amd import adjust
    from tests import oracle
    rows, cols, nbl, blocks, phased, _ = WORKLOADS[workload]
    lib = oracle.load()
    cores = os.cpu_count() or 1
    probe_n = 8192
    rng = np.random.default_rng(0)
    A = rng.standard_normal((probe_n, 64))
    M0 = np.asfortranarray(A @ A.T + np.eye(probe_n) * probe_n)
    del A

    def probe(t):
        lib.orc_set_threads(t)
        M = M0.copy(order="F")
        t0 = time.perf_counter()
        lib.orc_potrf_lower(probe_n, M.ctypes.data_as(oracle.f64p), probe_n)
        lib.orc_potri_lower(probe_n, M.ctypes.data_as(oracle.f64p), probe_n)
        return time.perf_counter() - t0

    # thread counts from few to many; a library's sweep stops once two larger counts in a row were slower than its best (a
    # container's CPU quota, not its visible core count, sets the optimum: on the pool's 256-"core" hosts 16 threads win)
    cand = sorted({c for c in (cores, cores // 2, cores // 4, 96, 64, 48, 32, 16, 8) if 1 <= c <= cores})
    libs = [("MKL runtime (libmkl_rt)", oracle.MKL), ("OpenBLAS (scipy wheel)", oracle.scipy_openblas_path())]
    only = os.environ.get("DNAGPU_CPU_LAPACK")            # diagnostic: "mkl" / "openblas" pins the library
    if only:
        libs = [l for l in libs if only.lower() in l[0].lower()]
    probes, best = {}, None
    for name, path in libs:
        if not path or not oracle.use_lapack(path):
            continue
        probe(cand[0])                                   # (first call: thread pool start-up)
        lib_best, worse = None, 0
        for t in cand:
            dt = probe(t)
            probes[f"{name} x{t}"] = round(probe_n ** 3 / dt / 1e12, 3)
            if best is None or dt < best[2]:
                best = (name, path, dt, t)
            if lib_best is None or dt < lib_best:
                lib_best, worse = dt, 0
            else:
                worse += 1
                if worse >= 2:
                    break
    del M0
    if best is None:
        oracle.use_lapack(None)
        name, path, threads, probe_rate = "built-in scalar Cholesky", None, 1, None
    else:
        name, path, _, threads = best
        probe_rate = probe_n ** 3 / best[2]
        oracle.use_lapack(path)
        lib.orc_set_threads(threads)
    d = tempfile.mkdtemp(prefix="dnagpu_cpu_")
    synth_kw = {}
    if "rows_hi" in WORKLOAD_KW.get(workload, {}) and path:
        # a small-block segmentation is cheap enough for the CPU to run WHOLE (sum n^3 ~ 1e13 per iteration): no strip, no extrapolation in n
        nb_s, rows_s, cols_s, synth_kw = 1, rows, cols, WORKLOAD_KW[workload]
    elif phased:
        nb_s = 3 if path else 2
        rows_s = max(4, (rows // blocks) * nb_s)
        cols_s = cols if path else min(cols, 40)
    else:
        nb_s, rows_s, cols_s = 1, rows, cols
        if not path:
            rows_s, cols_s = min(rows, 24), min(cols, 24)
    info = adjust.write_synthetic_network(d, "cpu", rows_s, cols_s, 0 if not synth_kw else nbl, nb_s, **synth_kw)
    if synth_kw:
        nb_s = info["blocks"]

    def new_adjustment():
        net = oracle.Network(os.path.join(d, "cpu"), phased)
        o = oracle.Adjustment(net, phased, threads=threads if path else 0)
        o.prepare()
        return o

    schedule, dt, n3, solves, note, mt_rate = "sequential", None, 0.0, 0, "", None
    if phased and path and threads >= 2 and lib.orc_set_threads_local(0) == 0:
        # the reference's multi-thread schedule: forward || reverse + combination
        f, r = new_adjustment(), new_adjustment()
        if lib.orc_adjust_forward_pass(r.h):               # (untimed: r must stand where a reverse pass starts)
            raise RuntimeError(lib.orc_adjust_error(r.h).decode())
        s0, n0 = r.solve_stats()
        half = max(1, threads // 2)
        errs = []

        def run(fn, h):
            lib.orc_set_threads_local(half)
            if fn(h):
                errs.append(lib.orc_adjust_error(h).decode())

        tf = threading.Thread(target=run, args=(lib.orc_adjust_forward_pass, f.h))
        tr = threading.Thread(target=run, args=(lib.orc_adjust_reverse_pass, r.h))
        t0 = time.perf_counter()
        tf.start(); tr.start(); tf.join(); tr.join()
        dt_mt = time.perf_counter() - t0
        if errs:
            raise RuntimeError(errs[0])
        sf, nf = f.solve_stats()
        sr, nr = r.solve_stats()
        solves_mt, n3_mt = sf + (sr - s0), nf + (nr - n0)
        f.close(); r.close()
        schedule, dt, n3, solves = "multi-thread", dt_mt, n3_mt, solves_mt
        mt_rate = n3_mt / dt_mt
        note = f"forward || reverse+combination on two threads x {half} LAPACK threads"
    elif phased and path and threads >= 2:
        # no per-thread control (OpenBLAS): the two passes as two pinned processes
        try:
            dt_mt, n3_mt, solves_mt, note = _cpu_multi_thread_by_processes(d, path, threads)
            schedule, dt, n3, solves = "multi-thread", dt_mt, n3_mt, solves_mt
            mt_rate = n3_mt / dt_mt
        except Exception as e:      # noqa: BLE001  (the sequential schedule below is measured either way)
            note = f"multi-thread schedule not measured ({e}); "
    # the sequential schedule on all tuned threads (the only one without per-thread LAPACK control)
    o = new_adjustment()
    lib.orc_set_threads(threads) if path else None
    t0 = time.perf_counter()
    o.iteration()
    dt_seq = time.perf_counter() - t0
    s_seq, n3_seq = o.solve_stats()
    o.close()
    seq_rate = n3_seq / dt_seq
    mt_note = note
    if dt is None or n3 / dt < seq_rate:
        note = (f"multi-thread schedule measured slower ({n3 / dt / 1e12:.3f} TFLOP/s: {mt_note}) than " if dt else mt_note) + f"sequential passes on {threads} LAPACK threads"
        schedule, dt, n3, solves = "sequential", dt_seq, n3_seq, s_seq
    oracle.use_lapack(None)
    # the combine pool (dnaadjust-multi.cpp:644-710: the combination solves of different blocks on hardware_concurrency() workers) as a
    # throughput bound: K Solve()-sized inversions at once, K x t = the tuned thread count; an upper bound on every schedule of the
    # reference on this host, so the best of the three is what the GPU is compared with
    limits = _cpu_limits()
    n_blk = int(round((n3_seq / max(1, s_seq)) ** (1.0 / 3.0)))
    pool_rates, pool_note = {}, None
    if phased and path and threads >= 4 and not os.environ.get("DNAGPU_CPU_NO_POOL"):
        for pool in (2, 4):
            if threads // pool < 2:
                continue
            try:
                pool_rates[f"{pool} x {threads // pool} threads"] = round(_cpu_combine_pool(d, path, threads, n_blk, pool), 3)
            except Exception as e:      # noqa: BLE001
                pool_note = f"combine pool of {pool} not measured ({e})"
                break
    strip_schedule, cpu_flops, pool_text = schedule, n3 / dt, ""
    if pool_rates:
        best_pool = max(pool_rates, key=pool_rates.get)
        if pool_rates[best_pool] * 1e12 > cpu_flops:
            schedule = "combine pool (throughput bound)"
            cpu_flops = pool_rates[best_pool] * 1e12
            pool_text = (f"; `value` prices the workload at the combine pool's throughput bound ({best_pool}: concurrent Solve()-sized dpotrf + dpotri of order "
                         f"{n_blk} as pinned processes, the chains' dependencies ignored: {cpu_flops / 1e12:.3f} TFLOP/s), which beat the strip's schedules")
    projected = sum_n3_per_step / cpu_flops
    return {
        "value": stations / projected,
        "unit": "stations/s",
        "cores": threads,
        "cores_visible": cores,
        "host_limits": limits,
        "kind": "port",
        "schedule": schedule,
        "lapack": f"{name}, {threads} threads (best of the dpotrf+dpotri probe at n = {probe_n})",
        "lapack_probe_tflops": probes,
        "probe_tflops_at_choice": None if probe_rate is None else probe_rate / 1e12,
        "sample": (f"one iteration of the CPU restatement over a {nb_s}-block, {info['stations']}-station strip of the workload's grid at the workload's "
                   f"block size (n ~ {n_blk} per Solve(), {solves} Solve() calls, sum n^3 = {n3:.3e}) in {dt:.2f} s = {n3 / dt / 1e12:.3f} TFLOP/s "
                   f"reference-equivalent, schedule: {strip_schedule} ({note}); sequential: {seq_rate / 1e12:.3f} TFLOP/s{pool_text}; extrapolated linearly in sum n^3 "
                   f"to the workload's {solves_per_step} Solve() calls per step"),
        "seconds_sample": dt,
        "tflops_reference_equivalent": cpu_flops / 1e12,
        "tflops_sequential_schedule": seq_rate / 1e12,
        "tflops_multi_thread_schedule": None if mt_rate is None else mt_rate / 1e12,
        "tflops_combine_pool": pool_rates or pool_note,
        "multi_thread_schedule": mt_note or None,
        # the same restatement run over the WHOLE workload once (committed record: not re-run here, it takes minutes to hours)
        "full_run_record": _full_run_record(workload),
    }


def _solves_of_reference_schedule(a, lib, its):
    """Solve() calls (and their n^3) the reference's forward / reverse / combination schedule would have made"""
    solves, ref = 0, 0.0
    for k in range(a.blockCount()):
        f, l, i = C.c_int(), C.c_int(), C.c_int()
        lib.dnaadj_block_flags(a.h, k, C.byref(f), C.byref(l), C.byref(i))
        m = 1 + (0 if i.value else 1) + (0 if (f.value or l.value or i.value) else 1)
        n3 = (3.0 * lib.dnaadj_block_station_count(a.h, k)) ** 3
        solves += its * m
        ref += its * m * n3
    return solves, ref


def _multi_gpu_line(args, world, dt, stations, B, its, solves, ref, per_rank, owners, check, driver, transport, rccl_ranks, multi_thread):
    """the JSON line of an N > 1 run from the per-rank records (alg, gemm_ms, issued, solves, completions, eliminations, exchange_ms,
    chain_ms, bytes), whichever way the ranks were started"""
    condensed = any(v["eliminations"] for v in per_rank)
    alg = sum(v["alg"] for v in per_rank)
    busiest = max(range(world), key=lambda r: per_rank[r]["gemm_ms"])
    gemm_ms = per_rank[busiest]["gemm_ms"]
    achieved = (per_rank[busiest]["alg"] / 1e12) / (gemm_ms / args.steps / 1e3) if gemm_ms > 0 else 0.0
    chains = (os.environ.get("DNAGPU_CHAINS", "4") + " chains") if multi_thread else "one chain"
    par = (f"condensed schedule inside dna_adjust::AdjustNetwork (C++): {B} blocks in contiguous runs over {world} ranks ({chains} per GPU), "
           "condensed blocks broadcast in place by ncclBroadcast, chains on the condensed blocks on every rank, coordinates by one ncclAllReduce"
           ) if condensed else (
           f"reference schedule inside dna_adjust::AdjustNetwork (C++): forward chain on rank 0, reverse chain on rank 1, combination solves "
           f"round-robin over {world} ranks; junction matrices by ncclSend / ncclRecv")
    return {
        "metric": "stations adjusted/sec + Cholesky TFLOP/s, phased adjustment, 1/2/4/8 MI355X",
        "value": stations * args.steps / dt, "unit": "stations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"stations": stations, "blocks": B, "iterations_to_converge": its, "mode": "phased", "solves_per_step": solves,
                   "schur_carry": condensed, "keep_factors": any(v["completions"] for v in per_rank),
                   "variance_matrices": "after the last iteration" if (not args.variances_every_iteration and any(v["completions"] for v in per_rank)) else "every iteration", "parallelism": par,
                   # who drove the ranks, what carried the exchange, and how many ranks the RCCL communicator itself counts (ncclCommCount; 0 = not RCCL)
                   "driver": driver, "transport": transport, "rccl_ranks": rccl_ranks,
                   "variance_propagation_in_step": bool(args.variance_propagation), "blocks_per_rank": [owners.count(r) for r in range(world)]},
        "cholesky_tflops": (alg / 1e12) / (dt / args.steps),
        "reference_equivalent_tflops": (ref / 1e12) / (dt / args.steps),
        # `achieved` / `frac`: the whole job's algorithmic flops over the whole step, per GPU (what the driver's clock supports: exchange, chains and
        # imbalance included); `*_gemm_busy`: the busiest rank's flops over the HIP-event time of its GEMM launches (the kernel's own rate)
        "roofline": {"kernel": "gemm_f64_dma_kernel (v_mfma_f64_16x16x4_f64 tile GEMM behind potrf/trtri/lauum)", "bound": "mfma",
                     "achieved": (alg / 1e12) / (dt / args.steps) / world, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s per GPU",
                     "frac": (alg / 1e12) / (dt / args.steps) / world / FP64_MFMA_PEAK_TFLOPS,
                     "achieved_gemm_busy": achieved, "frac_gemm_busy": achieved / FP64_MFMA_PEAK_TFLOPS,
                     "traffic": None, "rank": busiest, "gemm_ms_per_step": gemm_ms / args.steps,
                     "note": "frac: whole job, end to end, per GPU; *_gemm_busy: busiest rank, algorithmic flops of its steps / HIP-event time of its GEMM launches"},
        # host-side time of the last timed step, per rank: the exchange steps (broadcasts, all-reduce, their waits) and the chains on
        # the condensed blocks -- the serial remainder of the condensed schedule
        "exchange": {"exchange_ms_per_step": [v["exchange_ms"] for v in per_rank], "chain_phase_ms_per_step": [v["chain_ms"] for v in per_rank],
                     "payload_bytes_per_rank_per_step": [v["bytes"] for v in per_rank]},
        "check": check,
    }


def _check_block(a, folder, name, stations):
    import numpy as np
    try:
        a.GenerateStatistics()
        truth = np.fromfile(os.path.join(folder, name + ".truth"), dtype=np.float64).reshape(-1, 3)
        xyz = a.adjusted_coordinates(stations)
        return {"sigma_zero": a.GetSigmaZero(), "degrees_of_freedom": a.GetDegreesOfFreedom(),
                "max_abs_error_vs_truth_m": float(np.abs(xyz - truth).max()), "global_test": int(a.GetTestResult()),
                "chi_squared_limits": [a.GetChiSquaredLowerLimit(), a.GetChiSquaredUpperLimit()]}
    except Exception as e:                       # diagnostic only
        return {"error": str(e)}


HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
HBM_KINDS = ["unpermute_kernel (completed inverse back to the block's unknown order: np^2 read + written)",
             "init_ordered + form_normals_ordered (normals formed in the elimination's order: lower triangle written once)",
             "gemv_* substitution from the kept factor (factor read twice per solve)",
             "symv_* corrections = N^-1 rhs (n x np read once)",
             "pack_lower (staged store: n^2/2 read + written)"]


def _hbm_roofline(lib, ctx):
    """HBM-bound kernels of the step, measured in the one-chain step (nothing else on the device): algorithmic bytes per launch /
    HIP-event duration on the launch stream, against the 8 TB/s HBM3E peak (dnagpu_profile_hbm_get)"""
    b, ms, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_uint64 * 8)()
    if lib.dnagpu_profile_hbm_get(ctx, b, ms, n, 1) != 0:
        return None
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "measured_in": "the one-chain step (HIP events on the launch stream)", "kernels": []}
    for k, name in enumerate(HBM_KINDS):
        if n[k] == 0 or ms[k] <= 0:
            continue
        gbs = b[k] / 1e9 / (ms[k] / 1e3)
        out["kernels"].append({"kernel": name, "launches": int(n[k]), "algorithmic_gb_per_launch": round(b[k] / 1e9 / n[k], 3),
                               "avg_ms": round(ms[k] / n[k], 4), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 3)})
    return out


def _hbm(lib, ctx):
    free, total = C.c_size_t(), C.c_size_t()
    lib.dnagpu_mem_info(ctx, C.byref(free), C.byref(total))
    return free.value, total.value


def _hbm_report(lib, ctxs, after_prepare, staged):
    """the HBM budget of every rank: free after PrepareAdjustment (blocks, measurements, kept-factor budget decided) and at the end of the timed steps
    (chain workspaces, rigorous variance matrices or their staging buffers allocated); `staged`: the variance matrices live in page-locked host memory"""
    end = [_hbm(lib, c) for c in ctxs]
    return {"total_gb": [round(t / 1e9, 1) for _, t in end], "free_after_prepare_gb": [round(f / 1e9, 1) for f, _ in after_prepare],
            "free_at_end_gb": [round(f / 1e9, 1) for f, _ in end], "variances_staged_in_host_memory": staged}


def bench_one_process(folder, name, phased, args, devices, transport):
    """bench.py --gpus N without a launcher (WORLD_SIZE unset): ONE process drives the N GPUs, the mode dnaadjustwrapper linked to
    libdnagpu.so gets (a.devices -> one dna_adjust instance and one host thread per GPU inside the library, RCCL communicators made
    by the threads; dna_adjust_dist.cpp "one process, several GPUs").  The reference's parallel driver starts its threads inside the
    class the same way (dnaadjust-multi.cpp:92-244)."""
    import torch
    from dynadjust_amd import adjust
    if not phased:
        raise SystemExit("the simultaneous adjustment has one block: nothing to spread over GPUs by block; run it with --gpus 1")
    world = len(devices)
    a = adjust.DnaAdjust()
    multi_thread = bool(int(os.environ.get("DNAGPU_MULTI_THREAD", "1")))
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, devices=devices, dist_transport=transport, multi_thread=multi_thread,
                               schur_carry=not args.reference_schedule, keep_factors=not args.no_keep_factors, stage=args.stage,
                               defer_variances=0 if args.variances_every_iteration else int(os.environ.get("DNAGPU_DEFER_VARIANCES", "2")),
                               dist_two_level=bool(int(os.environ.get("DNAGPU_TWO_LEVEL", "1"))), reuse_factors=not args.no_reuse_factors,
                               chain_runs=args.chain_runs)
    a.PrepareAdjustment(p)
    lib = a.lib
    ctxs = [a.device_instance_context(r) for r in range(world)]
    hbm = [_hbm(lib, c) for c in ctxs]           # per rank, right after PrepareAdjustment: what the rank's blocks and workspaces left free

    def sync_all():
        for c in ctxs:
            lib.dnagpu_sync(c)
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def one_step():
        a.ResetAdjustment()
        st = a.AdjustNetworkDistributed()
        if st != adjust.ADJUST_SUCCESS:
            raise SystemExit(f"adjustment did not converge (status {st})")
        if args.variance_propagation:
            a.GenerateStatistics()

    for _ in range(args.warmup):
        one_step()
    for c in ctxs:
        lib.dnagpu_profile_enable(c, 0 if args.no_gemm_events else 1)
        lib.dnagpu_profile_reset(c)
    bytes0 = [a.device_instance_stats(r)["exchanged_bytes"] for r in range(world)]
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync_all()
    dt = time.perf_counter() - t0
    per_rank = []
    for r, c in enumerate(ctxs):
        prof_f, prof_ms, prof_n = C.c_double(), C.c_double(), C.c_uint64()
        lib.dnagpu_profile_get(c, C.byref(prof_f), C.byref(prof_ms), C.byref(prof_n))
        lib.dnagpu_profile_enable(c, 0)
        st = a.device_instance_stats(r)
        per_rank.append({"alg": st["algorithmic_flops"], "gemm_ms": prof_ms.value, "issued": prof_f.value, "solves": st["solves"],
                         "completions": st["completions"], "eliminations": st["eliminations"], "exchange_ms": st["exchange_ms"],
                         "chain_ms": st["chain_ms"], "bytes": (st["exchanged_bytes"] - bytes0[r]) / max(1, args.steps)})
    its = a.CurrentIteration()
    stations = lib.dnaadj_station_count(a.h)
    B = a.blockCount()
    owners = [a.block_owner(k) for k in range(B)]
    solves, ref = _solves_of_reference_schedule(a, lib, its)
    st0 = a.device_instance_stats(0)
    _, _, tr = a.dist_info()
    out = _multi_gpu_line(args, world, dt, stations, B, its, solves, ref, per_rank, owners, _check_block(a, folder, name, stations),
                          "C++ (libdnagpu.so): one process, one dna_adjust instance + host thread per GPU (a.devices)", tr, st0["rccl_ranks"], multi_thread)
    out["config"]["devices"] = list(devices)
    out["hbm_per_rank"] = _hbm_report(lib, ctxs, hbm, bool(lib.dnaadj_staged(a.h)))
    if len(set(devices)) < world:
        out["config"]["ranks_share_gpus"] = True     # DNAGPU_BENCH_SHARE_GPU=1: a code-path check on a box with fewer GPUs, NOT a scaling measurement
        out["n_gpus"] = len(set(devices))
        out["config"]["ranks"] = world
    a.close()
    return out


def bench_distributed_native(folder, name, phased, args, dist, rank, world, local_rank):
    """bench.py --gpus N (N > 1): one network, its blocks spread over N ranks by the library itself (strong scaling)."""
    import numpy as np
    import torch
    from dynadjust_amd import adjust
    if not phased:
        raise SystemExit("the simultaneous adjustment does not shard: run it with --gpus 1")
    a = adjust.DnaAdjust()
    shared = os.environ.get("DNAGPU_DIST_TRANSPORT") == "shared"
    fallback_note = None
    if not shared:
        # rank 0's ncclUniqueId to everybody through the control plane
        idt = torch.zeros(128, dtype=torch.uint8)
        err = ""
        try:
            if rank == 0:
                idt = torch.frombuffer(bytearray(adjust.rccl_unique_id()), dtype=torch.uint8).clone()
        except Exception as e:      # (librccl missing on rank 0: the others must not wait for an id that never comes)
            err = str(e)
        dist.broadcast(idt, src=0)
        try:
            if not err:
                a.attach_rccl(rank, world, bytes(idt.numpy().tobytes()), local_rank)
        except Exception as e:
            err = str(e)
        # A communicator that cannot start on ANY rank (no fabric path, a driver without dmabuf IPC, ...) must not cost the run: every rank
        # learns of it here, BEFORE the first collective, and all of them take the library's host-staged transport instead -- slower exchanges
        # (the two-level chains move a few MB per iteration), the same C++ driver, and the JSON line says so (config.transport, config.rccl_failed).
        bad = torch.tensor([1 if err else 0], dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            msgs = [None] * world
            dist.all_gather_object(msgs, err)
            fallback_note = next((m for m in msgs if m), "RCCL could not start")
            print(f"[bench] rank {rank}: RCCL could not start ({fallback_note}): host-staged transport between the ranks", file=sys.stderr, flush=True)
            a.close()
            a = adjust.DnaAdjust()
            shared = True
    # (DNAGPU_DIST_TRANSPORT=shared: launched ranks WITHOUT RCCL between them -- several processes on one GPU with DNAGPU_BENCH_SHARE_GPU=1: the
    #  launcher's command line, rendezvous, agreement and reporting paths on a one-GPU box; the library makes its own connections, host-staged)
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode, device=local_rank, dist_rank=rank, dist_world=world,
                               dist_transport="shared" if shared else None,
                               multi_thread=bool(int(os.environ.get("DNAGPU_MULTI_THREAD", "1"))),
                               schur_carry=not args.reference_schedule, keep_factors=not args.no_keep_factors, stage=args.stage, defer_variances=0 if args.variances_every_iteration else int(os.environ.get("DNAGPU_DEFER_VARIANCES", "2")),
                               dist_two_level=bool(int(os.environ.get("DNAGPU_TWO_LEVEL", "1"))), reuse_factors=not args.no_reuse_factors)
    a.PrepareAdjustment(p)
    lib, ctx = a.lib, a.device_context()
    hbm0 = _hbm(lib, ctx)

    def one_step():
        a.ResetAdjustment()
        st = a.AdjustNetworkDistributed()
        if st != adjust.ADJUST_SUCCESS:
            raise SystemExit(f"adjustment did not converge (status {st})")
        if args.variance_propagation:
            a.GenerateStatistics()

    for _ in range(args.warmup):
        t_s = time.perf_counter()
        one_step()
        print(f"[bench] warm-up step {time.perf_counter() - t_s:.2f} s", file=sys.stderr, flush=True)
    lib.dnagpu_profile_enable(ctx, 0 if args.no_gemm_events else 1)
    lib.dnagpu_profile_reset(ctx)
    bytes0 = a.exchange_stats()["bytes"]
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    lib.dnagpu_sync(ctx)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    prof_f, prof_ms, prof_n = C.c_double(), C.c_double(), C.c_uint64()
    lib.dnagpu_profile_get(ctx, C.byref(prof_f), C.byref(prof_ms), C.byref(prof_n))
    lib.dnagpu_profile_enable(ctx, 0)
    its = a.CurrentIteration()
    ex = a.exchange_stats()
    mine = {"alg": a.algorithmic_flops(), "gemm_ms": prof_ms.value, "issued": prof_f.value, "solves": a.solve_count(), "completions": a.completion_count(),
            "eliminations": a.elimination_count(), "exchange_ms": ex["exchange_ms"], "chain_ms": ex["chain_ms"], "bytes": (ex["bytes"] - bytes0) / max(1, args.steps),
            "rccl_ranks": a.device_instance_stats(0)["rccl_ranks"], "hbm0": hbm0, "hbm1": _hbm(lib, ctx), "staged": bool(lib.dnaadj_staged(a.h)),
            "batched_block_steps": a.batched_block_steps(), "batched_flops": a.batched_flops(), "plan": a.memory_plan(), "device": local_rank}
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    stations = lib.dnaadj_station_count(a.h)
    B = a.blockCount()
    owners = [a.block_owner(k) for k in range(B)]
    # statistics across the ranks (collective inside the library) and the distance from the truth the generator kept
    check = _check_block(a, folder, name, stations)
    out = None
    if rank == 0:
        solves, ref = _solves_of_reference_schedule(a, lib, its)
        _, _, tr = a.dist_info()
        out = _multi_gpu_line(args, world, dt, stations, B, its, solves, ref, allv, owners, check,
                              "C++ (libdnagpu.so): one process per GPU (torchrun ranks), RCCL called from the library", tr, min(v["rccl_ranks"] for v in allv),
                              p.multi_thread)
        if fallback_note:
            out["config"]["rccl_failed"] = fallback_note[:300]
        if len({v["device"] for v in allv}) < world:
            out["config"]["ranks_share_gpus"] = True     # DNAGPU_BENCH_SHARE_GPU=1: a code-path check on a box with fewer GPUs, NOT a scaling measurement
            out["n_gpus"] = len({v["device"] for v in allv})
            out["config"]["ranks"] = world
        out["hbm_per_rank"] = {"total_gb": [round(v["hbm1"][1] / 1e9, 1) for v in allv], "free_after_prepare_gb": [round(v["hbm0"][0] / 1e9, 1) for v in allv],
                               "free_at_end_gb": [round(v["hbm1"][0] / 1e9, 1) for v in allv], "variances_staged_in_host_memory": any(v["staged"] for v in allv)}
        # what every rank batched of its own blocks (block steps and share of its flops in the LAST step) and the plan PrepareAdjustment made on it
        out["batches_per_rank"] = {"batched_block_steps": [v["batched_block_steps"] for v in allv],
                                   "batched_fraction_of_flops": [round(v["batched_flops"] / v["alg"], 3) if v["alg"] else 0.0 for v in allv],
                                   "batch_members_beyond_first": [v["plan"].get("batch_members_beyond_first") for v in allv],
                                   "blocks_keeping_their_factor": [v["plan"].get("blocks_keeping_their_factor") for v in allv]}
    a.close()
    return out


def _so_hash():
    """first 16 hex digits of the SHA-256 of the libdnagpu.so this run loaded (the GPU box runs the binary built in the build container)"""
    import hashlib
    try:
        with open(os.path.join(ROOT, "dynadjust_amd", "libdnagpu.so"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def emit(result):
    """the ONE JSON line, last on stdout: whatever C libraries left in their stdio buffers (RCCL's version banner is printed through
    C stdio and would otherwise surface at exit, after the line) goes out first"""
    try:
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("DNAGPU_WORKLOAD", "cfg3"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reuse-inverses", action="store_true",
                    help="phased GNSS-only networks: keep the block inverses of the first iteration in HBM and reuse them "
                         "(identical results, half the Solve() calls; NOT what the reference does in phased mode, so not the default)")
    ap.add_argument("--stage", action="store_true",
                    help="the reference's --staged-adjustment: rigorous variance matrices leave HBM for page-locked host memory "
                         "(switches itself on when they do not fit; this flag forces it for measurements)")
    ap.add_argument("--reference-schedule", action="store_true",
                    help="every forward / reverse step inverts its block like the reference's Solve() (a.schur_carry = 0) instead of "
                         "eliminating the inner unknowns of the steps that are only carried on")
    ap.add_argument("--no-keep-factors", action="store_true",
                    help="condensed schedule without the retained factors (a.keep_factors = 0): every rigorous solve forms and inverts its block again")
    ap.add_argument("--variances-every-iteration", action="store_true",
                    help="a.defer_variances = 0: every rigorous solve forms its block's inverse like dna_adjust::Solve (default: the iterations "
                         "take their corrections from the completed factors and the rigorous variance matrices are formed once, after the last one)")
    ap.add_argument("--variance-propagation", action="store_true",
                    help="BASELINE.json configs[4]: the timed step also propagates the rigorous variances to every adjusted measurement "
                         "(GenerateStatistics: precisions of the adjusted measurements A S A^T from the resident variance matrices, chi-square, "
                         "sigma-zero, N-statistics) -- with `--workload cfg4` on 8 GPUs that is configs[4]")
    ap.add_argument("--plan", action="store_true",
                    help="dry run, no GPU needed: print PrepareAdjustment's plan for --gpus N GPUs of --plan-hbm-gb each (block owners, HBM budget per rank, "
                         "two-level runs, bytes of every exchange of an iteration: dnaadj_plan_distributed) and exit")
    ap.add_argument("--plan-hbm-gb", type=float, default=309.2, help="memory of one GPU for --plan (MI355X: 309.2 GB visible)")
    ap.add_argument("--no-one-chain", action="store_true", help="skip the extra one-chain step behind roofline.frac_one_chain")
    ap.add_argument("--chain-runs", type=int, default=-1,
                    help="a.chain_runs: the junction chains of a many-block network on one GPU cut into this many runs whose steps advance "
                         "together (lock-step chains); -1 = choose (32 runs from 512 blocks, 16 from 64), 0 = the chains step by step")
    ap.add_argument("--no-reuse-factors", action="store_true",
                    help="every iteration factors again (a.reuse_factors = 0): the schedule of rounds 1-4; by default iterations >= 2 of a GNSS-only "
                         "network keep the factors of iteration 1 and renew right-hand sides only")
    ap.add_argument("--no-refactor-leg", action="store_true", help="skip the extra step with a.reuse_factors = 0 behind `without_factor_reuse`")
    ap.add_argument("--no-gemm-events", action="store_true", help="diagnostic: no HIP events around the GEMM launches (roofline.achieved = 0)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-args", default="[]", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-role", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_role == "solve":
        _cpu_solve_role(*json.loads(args.cpu_args))
        return
    if args.cpu_baseline_role:
        _cpu_baseline_role(args.cpu_baseline_role, *json.loads(args.cpu_args))
        return
    if args.cpu_baseline_only:
        print(json.dumps(_cpu_baseline_sample(args.workload, *json.loads(args.cpu_args))), flush=True)
        return

    if args.plan:
        from dynadjust_amd import adjust
        rows, cols, nbl, blocks, phased, desc = WORKLOADS[args.workload]
        d = tempfile.mkdtemp(prefix="dnagpu_plan_")
        info = adjust.write_synthetic_network(d, "net", rows, cols, nbl, max(1, blocks), **WORKLOAD_KW.get(args.workload, {}))
        a = adjust.DnaAdjust()
        p = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode if phased else adjust.SimultaneousMode, multi_thread=phased,
                                   stage=phased and args.stage, schur_carry=not args.reference_schedule, keep_factors=not args.no_keep_factors)
        plan = a.plan_distributed(p, args.gpus, args.plan_hbm_gb * 1e9)
        a.close()
        plan["workload"] = desc
        plan["network"] = info
        print(json.dumps(plan), flush=True)
        return

    import torch
    launched = "WORLD_SIZE" in os.environ          # started by a launcher (torchrun: the driver's N > 1 runs), one process per GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the adjustment path has no CPU fallback")
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if launched and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # DNAGPU_BENCH_SHARE_GPU=1 (code-path checks on a box with fewer GPUs than ranks): ranks share devices, transport "local" inside the
    # library; the JSON line says so (`ranks_share_gpus`) and reports the devices actually used as n_gpus.  Never the default.
    share = bool(int(os.environ.get("DNAGPU_BENCH_SHARE_GPU", "0")))
    n_dev = torch.cuda.device_count()
    if args.gpus > n_dev and not share:
        raise SystemExit(f"--gpus {args.gpus}: this node shows {n_dev} GPU(s) (DNAGPU_BENCH_SHARE_GPU=1 lets ranks share devices for a code-path check)")
    # DNAGPU_DIST_BACKEND (launched runs): "native" (default) = control plane gloo, data path the library's own RCCL; "nccl" = control
    # plane torch's NCCL, data path still the library's RCCL; "gloo" = the Python TEST harness (tests/parallel_harness.py) with host
    # payloads and ranks that may share a GPU -- only when asked for, and the JSON line's config.driver says so.
    dist_backend = os.environ.get("DNAGPU_DIST_BACKEND", "native")
    if dist_backend == "gloo" or share:
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    print(f"libdnagpu.so sha256[:16] = {_so_hash()}", file=sys.stderr, flush=True)

    from dynadjust_amd import adjust
    from dynadjust_amd.device import DeviceContext  # noqa: F401  (fails loudly if the HIP library is missing)

    rows, cols, nbl, blocks, phased, desc = WORKLOADS[args.workload]
    d = tempfile.mkdtemp(prefix=f"dnagpu_bench_r{rank}_")
    t_w = time.perf_counter()
    copies = WORKLOAD_COPIES.get(args.workload, 1)
    if copies > 1:
        # several independently generated networks merged into one project (station / measurement indices offset, a network id per source)
        from tests import dnaformats
        parts = []
        for q in range(copies):
            info = adjust.write_synthetic_network(d, f"part{q}", rows, cols, nbl, max(1, blocks), seed=20260930 + q, **WORKLOAD_KW.get(args.workload, {}))
            parts.append(os.path.join(d, f"part{q}"))
        dnaformats.merge_networks(parts, os.path.join(d, "net"))
        with open(os.path.join(d, "net.truth"), "wb") as f:        # (the generator's true coordinates, in the merged station order)
            for part in parts:
                f.write(open(part + ".truth", "rb").read())
        info = dict(info, stations=info["stations"] * copies, baselines=info["baselines"] * copies, measurement_rows=info["measurement_rows"] * copies,
                    blocks=info["blocks"] * copies, networks=copies)
    else:
        info = adjust.write_synthetic_network(d, "net", rows, cols, nbl, max(1, blocks), **WORKLOAD_KW.get(args.workload, {}))
    blocks = info["blocks"]
    synth_s = time.perf_counter() - t_w
    stations = info["stations"]
    print(f"[bench] synthetic network written in {synth_s:.1f} s: {info}", file=sys.stderr, flush=True)

    if not launched and args.gpus > 1:
        # no launcher: this process drives all N GPUs itself (a.devices; RCCL between the library's per-GPU threads)
        devices = [r % n_dev for r in range(args.gpus)]
        result = bench_one_process(d, "net", phased, args, devices, "local" if len(set(devices)) < len(devices) else None)
        result["config"]["workload"] = desc
        result["config"]["libdnagpu_sha16"] = _so_hash()
        emit(result)
        return

    # DNAGPU_FORCE_DISTRIBUTED=1: run the N > 1 code path (RCCL communicator, device-resident payloads, collectives) with
    # however many ranks there are -- with one rank on a 1-GPU box it is the only way to exercise the RCCL transport there
    distributed = world > 1 or bool(int(os.environ.get("DNAGPU_FORCE_DISTRIBUTED", "0")))
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dist_backend in ("gloo", "native"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # N > 1: the driver is C++ (dna_adjust::AdjustPhasedDistributed), the exchange RCCL called from the library; torch.distributed
        # is the launcher's control plane here (unique-id broadcast, barriers, the max over the ranks' clocks).  If RCCL cannot start on
        # any rank, all ranks take the library's host-staged transport instead -- the same C++ driver, and never silently: config.transport
        # says "shared" and config.rccl_failed why (bench_distributed_native).
        if dist_backend == "gloo":
            from tests import parallel_harness
            result = parallel_harness.bench_distributed(d, "net", phased, args, dist, rank, world, local_rank, dist_backend)
            if rank == 0:
                result["config"]["driver"] = "Python TEST harness (tests/parallel_harness.py, DNAGPU_DIST_BACKEND=gloo): per-block C entry points, host payloads"
                result["config"]["transport"] = "gloo"
                result["config"]["rccl_ranks"] = 0
        else:
            result = bench_distributed_native(d, "net", phased, args, dist, rank, world, local_rank)
        if rank == 0:
            result["config"]["workload"] = desc
            result["config"]["libdnagpu_sha16"] = _so_hash()
            emit(result)
        dist.barrier()
        dist.destroy_process_group()
        return

    a = adjust.DnaAdjust()
    p = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode if phased else adjust.SimultaneousMode,
                               multi_thread=phased and bool(int(os.environ.get("DNAGPU_MULTI_THREAD", "1"))), device=local_rank,
                               reuse_inverses=phased and args.reuse_inverses, schur_carry=not args.reference_schedule, keep_factors=not args.no_keep_factors,
                               defer_variances=0 if args.variances_every_iteration else int(os.environ.get("DNAGPU_DEFER_VARIANCES", "2")),
                               stage=phased and args.stage, reuse_factors=not args.no_reuse_factors, chain_runs=args.chain_runs)
    t_p = time.perf_counter()
    a.PrepareAdjustment(p)
    prepare_s = time.perf_counter() - t_p
    lib = a.lib
    ctx = a.device_context()
    hbm_prepared = _hbm(lib, ctx)
    print(f"[bench] PrepareAdjustment {prepare_s:.1f} s (file load + host metadata + uploads), HBM free {hbm_prepared[0] / 1e9:.1f} of {hbm_prepared[1] / 1e9:.1f} GB",
          file=sys.stderr, flush=True)

    def one_step():
        a.ResetAdjustment()
        st = a.AdjustNetwork()
        if st != adjust.ADJUST_SUCCESS:
            raise SystemExit(f"adjustment did not converge (status {st})")
        if args.variance_propagation:
            a.GenerateStatistics()

    for _ in range(args.warmup):
        t_s = time.perf_counter()
        one_step()
        print(f"[bench] warm-up step {time.perf_counter() - t_s:.2f} s", file=sys.stderr, flush=True)
    lib.dnagpu_profile_enable(ctx, 0 if args.no_gemm_events else 1)
    lib.dnagpu_profile_reset(ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    lib.dnagpu_sync(ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof_f, prof_ms, prof_n = C.c_double(), C.c_double(), C.c_uint64()
    lib.dnagpu_profile_get(ctx, C.byref(prof_f), C.byref(prof_ms), C.byref(prof_n))
    lib.dnagpu_profile_enable(ctx, 0)

    iters = a.CurrentIteration()
    solves = a.solve_count()
    sum_n3 = a.solve_flops()           # reference-equivalent: n^3 per Solve() of the reference's schedule
    alg = a.algorithmic_flops()        # what the step executes: n^3 per inverse, the elimination's own count per carry-only step
    alg_min = a.minimal_work_flops()   # of that, the minimal schedule's share: every factorisation once, the variance matrices once (nothing re-done)
    elims = a.elimination_count()
    ms_per_step = dt * 1e3 / args.steps
    value = stations * args.steps / dt
    # roofline of the dominant kernel, the fp64 MFMA tile GEMM: algorithmic flops = n^3 per inverse (n^3/3 dpotrf + 2n^3/3
    # dpotri, the reference-equivalent count of a Solve()) and n_i^3/3 + n_i^2 n_j + n_i n_j^2 + n_j^3 per carry-only step
    # done by elimination, summed over the step, divided by the HIP-event duration of the gemm launches of the step
    gemm_ms_per_step = prof_ms.value / args.steps
    gemm_busy = (alg / 1e12) / (gemm_ms_per_step / 1e3) if gemm_ms_per_step > 0 else 0.0
    # `achieved` / `frac`: the step's algorithmic flops over the WHOLE step (the driver's clock supports it); `*_gemm_busy`: the same flops
    # over the time during which a GEMM was executing (union over the chains' streams, HIP events) -- the kernel's own rate, leaves and
    # launch gaps left out
    achieved = (alg / 1e12) / (ms_per_step / 1e3)
    out = {
        "metric": "stations adjusted/sec + Cholesky TFLOP/s, phased adjustment, 1/2/4/8 MI355X",
        "value": value,
        "unit": "stations/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": desc,
            "stations": stations, "baselines": info["baselines"], "measurement_rows": info["measurement_rows"], "blocks": blocks,
            "max_block_unknowns": info["max_block_unknowns"], "iterations_to_converge": iters, "solves_per_step": solves,
            "mode": "phased" if phased else "simultaneous", "reuse_inverses": bool(p.reuse_inverses),
            # a.reuse_factors: iterations >= 2 of a GNSS-only network keep the factors of iteration 1 (the reference's own rule in simultaneous mode,
            # dnaadjust.cpp:2452-2457); block steps / chain steps of a step that were served from a kept factor (right-hand sides only)
            "reuse_factors": bool(a.factor_reuses() or a.chain_step_reuses()), "chain_runs": a.chain_runs(), "factor_reuses_per_step": a.factor_reuses(),
            "chain_step_reuses_per_step": a.chain_step_reuses(),
            "schur_carry": bool(elims), "eliminations_per_step": elims, "keep_factors": bool(a.completion_count()),
            "variance_matrices": "after the last iteration" if (a.completion_count() and not args.variances_every_iteration and not args.reuse_inverses) else "every iteration",
            "libdnagpu_sha16": _so_hash(),
            "completions_per_step": a.completion_count(), "variance_propagation_in_step": bool(args.variance_propagation),
            # blocks of one shape through the large steps as one batch of merged launches (a.batch_blocks): block steps
            # (condensing, kept-block factorisation, variance matrices: up to 2 per block and iteration + 1 per block) that were batched
            "batched_block_steps_per_step": a.batched_block_steps(),
            "batched_fraction_of_flops": (a.batched_flops() / alg) if alg else 0.0,
            # outside the timed region (SURVEY section 8d excludes them from the metric), reported so that they cannot hide: writing the synthetic
            # .bst/.bms/.asl/.seg files, and PrepareAdjustment = reading them + host metadata (pair CSRs, appearance lists) + uploads
            "synth_write_s": round(synth_s, 2), "prepare_s": round(prepare_s, 2),
            # the GPU's memory: total, free after PrepareAdjustment (blocks, measurements), free after the timed steps (variance matrices,
            # kept factors, chain and batch workspaces allocated), and whether the variance matrices were staged to host memory
            "hbm_gb": {"total": round(hbm_prepared[1] / 1e9, 1), "free_after_prepare": round(hbm_prepared[0] / 1e9, 1),
                       "free_at_end": round(_hbm(lib, ctx)[0] / 1e9, 1), "variances_staged_in_host_memory": bool(lib.dnaadj_staged(a.h))},
            # PrepareAdjustment's memory plan and what the staged store moved: packed variance matrices in page-locked host memory / packed in
            # HBM past the host's memory limit (the container's cgroup, not /proc/meminfo), blocks that keep their factor between the
            # condensing step and the rigorous solve, bytes copied to the host in the LAST step and how long the host waited for them
            "memory_plan": a.memory_plan(),
            "parallelism": "1 GPU, one chain" if not p.multi_thread else
            "1 GPU, %s chains (multi_thread: the independent block steps of the condensed schedule are served by every chain; the two junction chains run side by side)" % os.environ.get("DNAGPU_CHAINS", "4"),
        },
        "cholesky_tflops": (alg / 1e12) / (ms_per_step / 1e3),
        # the same wall time priced at the reference's own work (n^3 for each of its Solve() calls): what a CPU or GPU
        # running the reference's schedule would have to sustain to be as fast; exceeds the hardware peak when steps are eliminated
        "reference_equivalent_tflops": (sum_n3 / 1e12) / (ms_per_step / 1e3),
        "roofline": {
            "kernel": "gemm_f64_dma_kernel (v_mfma_f64_16x16x4_f64 tile GEMM behind potrf/trtri/lauum)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "achieved_gemm_busy": gemm_busy,
            "frac_gemm_busy": gemm_busy / FP64_MFMA_PEAK_TFLOPS,
            "frac_end_to_end": achieved / FP64_MFMA_PEAK_TFLOPS,      # (= frac; the name earlier rounds used)
            # `frac` prices the flops the step EXECUTED; this one prices only those of the minimal schedule -- every factorisation once and the
            # variance matrices once -- over the same wall time, so that work done again (a factor made a second time where HBM denies a kept
            # one, an iteration that factors again) can never read as useful work
            "frac_min_work": (alg_min / 1e12) / (ms_per_step / 1e3) / FP64_MFMA_PEAK_TFLOPS,
            "algorithmic_flops_per_step": alg, "minimal_work_flops_per_step": alg_min,
            "frac_one_chain": None,           # filled below: the same step on ONE chain, timed in this run
            "traffic": traffic_from_profile(args.workload)[0],
            "traffic_source": traffic_from_profile(args.workload)[1],
            "traffic_unit": "bytes per launch (memory-side, FETCH_SIZE x 2 + WRITE_SIZE from the committed rocprofv3 --pmc passes of this workload)",
            "launches_per_step": prof_n.value / args.steps,       # tile-GEMM products per step
            "gemm_ms_per_step": gemm_ms_per_step,
            "issued_tflops": (prof_f.value / 1e12) / (prof_ms.value / 1e3) if prof_ms.value > 0 else 0.0,
        },
    }
    # outside the timed region: the adjusted network against what the synthetic measurements were drawn from -- size-independent
    # evidence, at the bench's full size, that the fast schedule adjusts correctly (sigma zero ~ 1: residuals match the stated
    # variances; every station within a few standard deviations of its true position; constrained corners unmoved)
    try:
        import numpy as np
        a.GenerateStatistics()
        truth = np.fromfile(os.path.join(d, "net.truth"), dtype=np.float64).reshape(-1, 3)
        xyz = a.adjusted_coordinates(stations)
        out["check"] = {"sigma_zero": a.GetSigmaZero(), "degrees_of_freedom": a.GetDegreesOfFreedom(),
                        "max_abs_error_vs_truth_m": float(np.abs(xyz - truth).max()), "global_test": int(a.GetTestResult()),
                        "chi_squared_limits": [a.GetChiSquaredLowerLimit(), a.GetChiSquaredUpperLimit()]}
    except Exception as e:                       # diagnostic only
        out["check"] = {"error": str(e)}
    a.close()
    # what the chains' overlap hides: the same step with ONE chain (nothing overlapped), one warm-up and one timed step, measured here
    if p.multi_thread and not args.no_one_chain:
        try:
            a1 = adjust.DnaAdjust()
            p1 = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode, multi_thread=False, device=local_rank, reuse_inverses=p.reuse_inverses,
                                        schur_carry=p.schur_carry, keep_factors=p.keep_factors, defer_variances=p.defer_variances, stage=p.stage,
                                        reuse_factors=p.reuse_factors)
            a1.PrepareAdjustment(p1)
            ctx1 = a1.device_context()
            for timed in (False, True):
                lib.dnagpu_profile_hbm_enable(ctx1, 1 if timed else 0)
                a1.ResetAdjustment()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                if a1.AdjustNetwork() != adjust.ADJUST_SUCCESS:
                    raise RuntimeError("the one-chain step did not converge")
                if args.variance_propagation:
                    a1.GenerateStatistics()
                lib.dnagpu_sync(a1.device_context())
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
            out["roofline"]["frac_one_chain"] = (a1.algorithmic_flops() / 1e12) / dt1 / FP64_MFMA_PEAK_TFLOPS
            out["roofline"]["ms_per_step_one_chain"] = dt1 * 1e3
            out["roofline_hbm"] = _hbm_roofline(lib, ctx1)
            a1.close()
        except Exception as e:                   # diagnostic only
            out["roofline"]["frac_one_chain_error"] = str(e)
    # the schedule of rounds 1-4 beside it: every iteration factors again (a.reuse_factors = 0) -- one warm-up and one timed step, measured here
    if phased and out["config"]["reuse_factors"] and not args.no_refactor_leg:
        try:
            a2 = adjust.DnaAdjust()
            p2 = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode, multi_thread=p.multi_thread, device=local_rank, reuse_inverses=p.reuse_inverses,
                                        schur_carry=p.schur_carry, keep_factors=p.keep_factors, defer_variances=p.defer_variances, stage=p.stage, reuse_factors=False)
            a2.PrepareAdjustment(p2)
            for timed in (False, True):
                a2.ResetAdjustment()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if a2.AdjustNetwork() != adjust.ADJUST_SUCCESS:
                    raise RuntimeError("the step without factor reuse did not converge")
                if args.variance_propagation:
                    a2.GenerateStatistics()
                lib.dnagpu_sync(a2.device_context())
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t2
            out["without_factor_reuse"] = {"ms_per_step": dt2 * 1e3, "value": stations / dt2, "unit": "stations/s",
                                           "frac": (a2.algorithmic_flops() / 1e12) / dt2 / FP64_MFMA_PEAK_TFLOPS,
                                           "frac_min_work": (a2.minimal_work_flops() / 1e12) / dt2 / FP64_MFMA_PEAK_TFLOPS,
                                           "iterations": a2.CurrentIteration()}
            a2.close()
        except Exception as e:                   # diagnostic only
            out["without_factor_reuse"] = {"error": str(e)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, iters, solves, sum_n3, stations)
    emit(out)


if __name__ == "__main__":
    main()
