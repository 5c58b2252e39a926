#!/usr/bin/env python3
"""Headline benchmark: accepted ODE steps / second, batched Brusselator N=32 (d=64), nu=3, dense EKF1,
dynamic calibration, fp64 (BASELINE.json metric).

A "step" of this benchmark is ONE batched attempted solver step: every trajectory of the batch runs
predict -> linearise -> calibrate -> correct -> error norm on the GPU, then the PI controller accepts /
rejects and proposes the next step sizes -- by default in a controller kernel on the device (SURVEY 8(f)-1,
`--controller device`: the K timed steps are enqueued back to back and synchronised once), or on the host
with one round trip per step (`--controller host`, the north star's division of labour; same decisions, see
tests/test_gpu_round2.py). Inputs are resident in HBM before the timed region.
`value` = accepted trajectory-steps of all ranks / wall time of the slowest rank.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
Multi-GPU: launched by torch.distributed.run, one rank per GPU. Default `--scaling strong`: the BASELINE
configuration, ONE batch of --batch (1024) trajectories sharded contiguously over the ranks
(`sharding.shard_bounds`: 128 per GPU at 8 GPUs), no data-path collective. `--scaling weak` keeps --batch
trajectories per GPU instead. torch.distributed (the driver's launch contract) carries the barrier and the
MAX/SUM of the timings; the path's own exchange, the final gather of per-trajectory results, is one
ncclAllGather through the ctypes binding of librccl.so in probdiffeq_hip/sharding.py.
"""

import argparse
import json
import os
import sys
import time

# the CPU baseline runs one trajectory per OpenMP thread with ONE BLAS thread each: keep OpenBLAS from spawning a pool per call
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

N_SPATIAL = 32
NU = 3
TOL = 1e-6
DT0 = 0.01
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak
HBM_PEAK_GBS = 8000.0
PMC_SUMMARIES = ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")  # newest committed rocprofv3 --pmc summary first


def problem_batch(B, offset=0, total=None):
    from probdiffeq_hip import taylor
    from probdiffeq_hip import vector_fields as V

    total = total or B
    vf = V.brusselator(N_SPATIAL)
    x = np.linspace(0.0, 1.0, N_SPATIAL)
    phases = 2.0 * np.pi * (offset + np.arange(B)) / total
    y0 = np.concatenate([1.0 + np.sin(2 * np.pi * x[None, :] + phases[:, None]), 3.0 * np.ones((B, N_SPATIAL))], axis=1)
    tcs = taylor.odejet_padded_scan(vf, [y0], num=NU)
    return vf, tcs


def algorithmic_flops_lq(D, d):
    """Row-Householder LQ of the stack [T | Mt], T triangular (D x D), Mt = (Abar P^-1 (x) I) Ls with the D - d columns of the
    rank-(D - d) state factor (r03: the noise-free measurement update leaves rank D - d): reflector i has
    1 + (D - d) entries and acts on the D - i rows below it -> 4 (D - d) D^2 / 2 = 2 D^2 (D - d). (r02 ran the same LQ on D
    columns: 2 D^3; SURVEY 8(d)'s (10/3) D^3 is the unstructured QR of the 2D x D stack.)"""
    return 2.0 * D * D * (D - d)


def _library_has_kernel(base_name):
    """Whether the built library contains a kernel of this (unmangled, template arguments stripped) name: the code object is
    embedded in the .so, the mangled kernel names contain the plain identifier."""
    global _LIB_BYTES
    if _LIB_BYTES is None:
        try:
            _LIB_BYTES = open(os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd", "csrc", "libpnode_hip.so"), "rb").read()
        except OSError:
            _LIB_BYTES = b""
    return base_name.encode() in _LIB_BYTES


_LIB_BYTES = None


def measured_traffic(kernel_prefix, batch):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/rNN_pmc_summary.json, produced on an MI355X by scripts/collect_profile.sh with separate
    --pmc passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), scaled to this batch.
    A summary is REFUSED (None) when one of the kernels it names is not in the built library: the profile then belongs to
    another version of the kernels and its traffic says nothing about this run."""
    data = None
    for name in PMC_SUMMARIES:
        try:
            data = json.load(open(os.path.join(ROOT, "profiles", name)))
            break
        except OSError:
            continue
    if data is None:
        return None
    stale = sorted({k.split("<")[0] for k in data if "hbm_bytes_per_launch" in data[k] and not _library_has_kernel(k.split("<")[0].split("::")[-1])})
    if stale:
        print(f"bench.py: profile names kernels the library does not contain ({', '.join(stale)}): traffic not reported", file=sys.stderr)
        return None
    total = 0.0
    for name, v in data.items():  # the LQ runs as two launches (first pass + remaining passes): sum them
        if name.startswith(kernel_prefix) and "hbm_bytes_per_launch" in v:
            total += v["hbm_bytes_per_launch"]
    return total * batch / 1024.0 if total > 0.0 else None  # the profile run uses 1024 trajectories


def _granted_cpus():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota where one is set."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        mx, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if mx != "max":
            quota = max(1, int(float(mx) / float(period)))
    except (OSError, ValueError):
        pass
    return aff, quota


def _cpu_workers(n, seconds, backend, cores):
    """n worker processes (oracle/cpu_baseline_worker.py), one trajectory and one pinned core each, started together."""
    import subprocess

    worker = os.path.join(ROOT, "oracle", "cpu_baseline_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(i), str(cores[i % len(cores)]) if cores else "-1", str(seconds), backend],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for i in range(n)]
    for p in procs:
        assert p.stdout.readline().strip() == "ready"
    for p in procs:
        p.stdin.write("go\n")
        p.stdin.flush()
    res = [json.loads(p.stdout.readline()) for p in procs]
    for p in procs:
        p.wait()
    return res


def cpu_baseline(seconds=5.0, repeats=2):
    """The oracle ("port") timed on this host's cores on a bounded sample of the same workload:
    oracle/pn_oracle_c.c (C restatement of oracle/pn_oracle.py, checked against it in tests/test_oracle_c.py) with its
    LAPACK / BLAS back end (blocked dgeqrf, dgemm, dtrsm of the OpenBLAS that scipy ships, one BLAS thread), ONE PROCESS PER
    GRANTED CORE, each pinned to its own core and running the adaptive loop of one trajectory of the batch for `seconds`; all
    processes run their timed region together; best of `repeats`. (r02 ran the trajectories as OpenMP threads of one process:
    OpenBLAS serialises concurrent callers on internal locks -- 599 accepted steps/s on one thread, 957 on sixteen -- while the
    plain-loop back end scaled linearly, 81 per thread: scripts/cpu_scaling_probe.py, profiles/r03_cpu_scaling_probe.txt.)"""
    aff, quota = _granted_cpus()
    threads = int(os.environ.get("PNODE_CPU_THREADS", min(aff, quota) if quota else aff))
    path = os.path.join(ROOT, "oracle", "libpn_oracle_c.so")
    vf, tcs = problem_batch(1, 0, 1024)
    if os.path.exists(path):
        allowed = sorted(os.sched_getaffinity(0))
        stride = max(1, len(allowed) // threads)
        cores = allowed[::stride][:threads]  # distinct cores spread over the mask (not SMT siblings where the mask allows)
        single = _cpu_workers(1, seconds, "lapack", cores[:1])[0]
        runs = []
        for _ in range(repeats):
            res = _cpu_workers(threads, seconds, "lapack", cores)
            runs.append((sum(r["accepted"] / r["seconds"] for r in res), sum(r["attempted"] / r["seconds"] for r in res)))
        best = max(runs)
        single_v = single["accepted"] / single["seconds"]
        return {"value": best[0], "unit": "accepted steps/s", "cores": int(threads), "kind": "port",
                "single_core_value": single_v, "per_core_value_all_busy": best[0] / threads,
                "sample": f"oracle/pn_oracle_c.c, {single['backend']}; {threads} processes, one trajectory of the same batch and one "
                          f"pinned core each, {seconds:.0f}s each, started together: best of {repeats} runs after a warm-up "
                          f"({', '.join(f'{r[0]:.0f}' for r in runs)} accepted steps/s; {best[1]:.0f} attempted steps/s); "
                          f"{best[0] / threads:.1f} steps/s per core with all cores busy, {single_v:.1f} on one core alone; "
                          f"affinity mask {aff} cpus, cgroup quota {quota}, os.cpu_count() {os.cpu_count()}"}
    # fallback: NumPy oracle, single process
    import pn_oracle as po
    from helpers import oracle_problem

    cfg = po.SolverConfig(nu=NU, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    acfg = po.AdaptiveConfig(TOL, TOL)
    prob = oracle_problem(vf)
    accepted = 0
    t0 = time.perf_counter()
    ast = po.adaptive_init(po.initial_state(cfg, [c[0] for c in tcs], 0.0), DT0)
    while time.perf_counter() - t0 < 3 * seconds:
        ast = po.rejection_loop(cfg, acfg, prob, ast, np.inf)
        accepted += 1
    el = time.perf_counter() - t0
    return {"value": accepted / el, "unit": "accepted steps/s", "cores": 1, "kind": "port",
            "sample": f"NumPy oracle (oracle/pn_oracle.py), 1 trajectory, {accepted} accepted steps in {el:.1f}s"}


def solve_on_handle(B, device_controller, device, repeats=3, t1=10.0):
    """The isotropic whole solve on an EXISTING handle (C-ABI calls of ivpsolve.solve_adaptive_save_at: pnode_set_initial,
    pnode_set_tolerances, the native driver, the read-back), fastest of `repeats`: creating a handle (device and pinned-host
    allocations) takes 5 to 40 ms depending on the box, more than the 25 ms solve it would be timed with."""
    from probdiffeq_hip import ivpsolve, ivpsolvers
    from probdiffeq_hip.impl import impl

    vf, tcs = problem_batch(B)
    impl.select("isotropic", ode_shape=(vf.d,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(NU), ivpsolvers.correction_ts0()))
    asolver = ivpsolve.adaptive(solver, atol=TOL, rtol=TOL)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, t1, 5)
    batch = ivpsolve._make_batch(vf, init, solver, len(save_at) - 1, device)
    best = None
    for _ in range(repeats):
        tic = time.perf_counter()
        batch.set_initial(init.tcoeffs, np.zeros(B), init.output_scale)
        batch.set_tolerances(TOL, TOL)
        steps, attempts, failed = batch.solve_adaptive_save_at(save_at, np.full(B, DT0), asolver.control, asolver.numerical_zero,
                                                               device_controller=device_controller)
        means = batch.read_checkpoint_means()
        el = time.perf_counter() - tic
        r = {"accepted": int(np.sum(steps[:, -1])), "attempted": int(np.sum(attempts)), "seconds": el,
             "accepted_steps_per_s": int(np.sum(steps[:, -1])) / el, "finite": bool(np.all(np.isfinite(means)) and not failed.any())}
        if best is None or el < best["seconds"]:
            best = r
    batch.close()
    best["timing"] = f"fastest of {repeats} solves on one handle (set_initial, native driver, read-back); handle creation excluded"
    return best


def whole_solve(model, B, device_controller, device):
    """One whole adaptive solve of the headline batch, t in [0, 10] (the reference's interval, ivps.py:124-156), through the public
    API: accepted / attempted steps and wall time including the controller and the checkpoint interpolation."""
    from probdiffeq_hip import ivpsolve, ivpsolvers
    from probdiffeq_hip.impl import impl

    vf, tcs = problem_batch(B)
    impl.select(model, ode_shape=(vf.d,))
    corr = ivpsolvers.correction_ts1() if model == "dense" else ivpsolvers.correction_ts0()
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(NU), corr))
    asolver = ivpsolve.adaptive(solver, atol=TOL, rtol=TOL)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, 10.0, 5)
    tic = time.perf_counter()
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=DT0, adaptive_solver=asolver, device=device,
                                          device_controller=device_controller)
    el = time.perf_counter() - tic
    acc, att = int(np.sum(sol.num_steps[..., -1])), int(np.sum(sol.num_attempts))
    return {"accepted": acc, "attempted": att, "seconds": el, "accepted_steps_per_s": acc / el, "finite": bool(np.all(np.isfinite(sol.u)))}


def extras(args, batch, loop, active, dense, device_ctl, device):
    """Legs outside the timed region (N = 1): the same batch stepped with the HOST controller (the north star's division of
    labour: one round trip per attempt), whole solves with both controllers (the figure that does not depend on the window of
    attempts), and the isotropic EKF0 nu = 3 line of SURVEY 8(d) with its fraction of the HBM roofline."""
    out = {}
    if device_ctl:
        k = min(args.steps, 20)
        t0 = time.perf_counter()
        for _ in range(k):
            loop.attempt(active)
        out["host_controller_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / k
    batch.close()
    B = args.batch
    ws = {}
    for name, dev in (("device_controller", True), ("host_controller", False)):
        ws[name] = whole_solve("dense" if dense else "isotropic", B, dev, device)
    out["whole_solve"] = {"t_span": [0.0, 10.0], "save_at_points": 5, **ws}
    if dense:
        n = NU + 1
        d = 2 * N_SPATIAL
        iso = {}
        for name, dev in (("device_controller", True), ("host_controller", False)):
            whole_solve("isotropic", B, dev, device)  # warm-up (code paging)
            r = solve_on_handle(B, dev, device)
            byts = 2 * 8 * (n * d + n * n)  # state in + out per attempted step
            r["algorithmic_bytes_per_step"] = byts
            r["hbm_GBps"] = r["attempted"] / r["seconds"] * byts / 1e9
            r["hbm_frac"] = r["hbm_GBps"] / HBM_PEAK_GBS
            if dev:
                # the device-resident controller keeps the state of a trajectory in the registers of its wavefront over the attempts of
                # a launch and writes only accepted states: the bytes it actually moves are half of SURVEY's figure, and the loop
                # is bound by the latency of one attempt (a serial chain per trajectory), not by HBM
                r["moved_bytes_per_accepted_step"] = byts // 2
                r["us_per_attempt_of_a_trajectory"] = r["seconds"] / (r["attempted"] / B) * 1e6
            iso[name] = r
        # the ceiling with the machine FILLED (1024 trajectories put one wavefront on each of the 1024 SIMDs): the same solve with
        # 8 and 64 wavefronts per SIMD's worth of trajectories, device controller; fraction of the HBM roofline on the bytes the
        # kernel actually moves (the accepted states: mean, factor, output scale)
        filled = {}
        moved = 8 * (n * d + n * n + 1)
        for Bf in (8192, 65536):
            r = solve_on_handle(Bf, True, device, repeats=2)
            r["moved_bytes_per_accepted_step"] = moved
            r["hbm_GBps_moved"] = r["accepted"] / r["seconds"] * moved / 1e9
            r["hbm_frac_moved"] = r["hbm_GBps_moved"] / HBM_PEAK_GBS
            r["hbm_frac_survey_formula"] = r["attempted"] / r["seconds"] * byts / 1e9 / HBM_PEAK_GBS
            r["us_per_attempt_of_a_trajectory"] = r["seconds"] / (r["attempted"] / Bf) * 1e6
            filled[f"batch_{Bf}"] = r
        out["isotropic_ekf0_nu3"] = {"workload": f"Brusselator N=32 (d=64), nu=3, isotropic EKF0 (the reference's own model for this PDE), "
                                                 f"filter, dynamic, tol 1e-6, batch {B}, whole solve t in [0, 10]", **iso,
                                     "machine_filled": filled}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024,
                    help="trajectories in the global batch (--scaling strong) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: ONE batch of --batch trajectories sharded over the GPUs (BASELINE config 4); weak: --batch per GPU")
    ap.add_argument("--lq", default="one-launch", choices=["one-launch", "two-launch"],
                    help="time update of the MFMA path: the library's default (one launch, overlapping 64-row passes) or the "
                         "two-launch kernels with 32-row passes (A/B comparison)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs outside the timed region (N = 1 only): host-controller attempts, whole solves of the "
                         "headline batch with both controllers, the isotropic EKF0 line")
    ap.add_argument("--controller", default="device", choices=["device", "host"],
                    help="device: accept/reject + PI law in a controller kernel between the attempts, the timed steps are enqueued "
                         "back to back with one synchronisation at the end (pnode_advance_device); host: one round trip per step")
    ap.add_argument("--model", default="dense", choices=["dense", "isotropic"])
    ap.add_argument("--diag-fixed-dt", type=float, default=0.0,
                    help="diagnostics only: constant step size, controller bypassed")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal knobs (1-GPU box): PNODE_BENCH_BACKEND=gloo runs the multi-rank branch over CPU tensors,
    # PNODE_BENCH_DEVICE=0 pins every rank to one GPU. The driver's real runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("PNODE_BENCH_BACKEND", "nccl")
    if "PNODE_BENCH_DEVICE" in os.environ:
        local = int(os.environ["PNODE_BENCH_DEVICE"])
    tdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist

        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from probdiffeq_hip import _backend, ivpsolve, ivpsolvers
    from probdiffeq_hip.impl import impl

    from probdiffeq_hip import sharding

    if args.scaling == "strong":
        total = args.batch
        lo, hi = sharding.shard_bounds(total, world, rank)
    else:
        total = args.batch * world
        lo, hi = rank * args.batch, (rank + 1) * args.batch
    B = hi - lo
    if B <= 0:
        raise SystemExit(f"rank {rank}: empty shard ({total} trajectories over {world} ranks)")
    vf, tcs = problem_batch(B, lo, total)
    d = vf.d
    D = (NU + 1) * d
    dense = args.model == "dense"
    impl.select(args.model, ode_shape=(d,))
    corr = ivpsolvers.correction_ts1() if dense else ivpsolvers.correction_ts0()
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(NU), corr))
    asolver = ivpsolve.adaptive(solver, atol=TOL, rtol=TOL)
    init = solver.initial_condition(tcs, 1.0)
    batch = ivpsolve._make_batch(vf, init, solver, 1, device=local)
    # time update of the MFMA path: the library default is the one-launch LQ (pnode_dense_lq4.hip); the kernel choice never
    # depends on the batch size (a shard of a batch must give the bits of the unsharded solve)
    one_launch_lq = dense and args.lq == "one-launch"
    if dense and not one_launch_lq:
        batch.set_generic_path(3)
    batch.set_initial(init.tcoeffs, np.zeros(B), init.output_scale)
    batch.set_tolerances(TOL, TOL)
    batch.set_phase_timing(True)  # HIP events around the step (and between the dense kernels)
    loop = ivpsolve._AdaptiveLoop(batch, asolver, np.zeros(B), DT0)
    loop.time_device = True
    active = np.ones(B, dtype=bool)

    def barrier():
        if dist is not None:
            import torch

            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()

    if args.diag_fixed_dt > 0.0:
        err_buf = np.zeros(B)

        def diag_attempt(_active):
            batch.attempt_step(np.full(B, args.diag_fixed_dt), np.ones(B, np.uint8), err_buf)
            loop.device_ms += batch.last_step_ms()
            batch.commit(np.ones(B, np.uint8))
            loop.steps += 1

        loop.attempt = diag_attempt
    phase = np.zeros(4)
    device_ctl = args.controller == "device" and args.diag_fixed_dt == 0.0
    if device_ctl:
        # controller on the device: W warm-up attempts, then EXACTLY K attempts of the whole batch enqueued back to back
        dt_state, lp_state = np.full(B, DT0), np.zeros(B)
        never = np.full(B, np.inf)
        ctl = asolver.control
        if args.warmup > 0:
            batch.advance_device(never, dt_state, lp_state, ctl, asolver.numerical_zero, args.warmup)
        barrier()
        acc0 = int(batch.read_time()[1].sum())
        t0 = time.perf_counter()
        status, attempts = batch.advance_device(never, dt_state, lp_state, ctl, asolver.numerical_zero, args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        assert np.all(attempts == args.steps), "a trajectory stopped attempting inside the timed region"
        accepted = int(batch.read_time()[1].sum()) - acc0
        dev_ms = batch.last_step_ms()  # HIP events around the K attempts on the handle's stream
        if dense:
            ph, cnt = batch.phase_ms_mean()  # HIP events around every kernel of EVERY timed attempt, on the handle's stream
            assert cnt == args.steps
            phase = np.array(ph) * args.steps
    else:
        for _ in range(args.warmup):
            loop.attempt(active)
        barrier()
        acc0 = int(loop.steps.sum())
        loop.device_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.attempt(active)
            if dense:
                phase += batch.last_phase_ms()
        barrier()
        elapsed = time.perf_counter() - t0
        accepted = int(loop.steps.sum()) - acc0
        dev_ms = loop.device_ms

    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([accepted], dtype=torch.float64, device=tdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        accepted = int(cnt.item())
        # the path's only exchange: gather per-trajectory results (here: current state u, [B, d]) -- one ncclAllGather over
        # RCCL through ctypes; the 128-byte ncclUniqueId travels over the process group the launch contract created
        u_local = batch.read_mean(0)[:, 0, :].copy()
        bmax = torch.tensor([B], dtype=torch.int64, device=tdev)
        dist.all_reduce(bmax, op=dist.ReduceOp.MAX)
        pad = np.zeros((int(bmax.item()), d))
        pad[:B] = u_local  # shards differ by at most one trajectory: pad to equal shapes
        # the path is agreed on by ALL ranks before anyone enters a collective (a rank that failed to create its communicator
        # must not sit in a torch collective while the others sit in ncclAllGather)
        comm, why = None, ""
        try:
            if backend != "nccl":
                raise RuntimeError("rehearsal backend")

            def via_group(raw):
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                return box[0]

            comm = sharding.RcclGather(rank, world, local, exchange_id=via_group)
        except Exception as e:
            why = f"{type(e).__name__}: {e}"
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        if int(ok.item()) == 1:
            gather_how = "rccl-ctypes"
            gathered = sharding.gather_results(pad, comm)  # (a failure here ends the run with a non-zero exit)
            comm.close()
        else:
            if comm is not None:
                comm.close()
            gather_how = "torch.distributed fallback on every rank (" + "; ".join(f"rank {i}: {w}" for i, w in enumerate(reasons) if w) + ")"
            t = torch.from_numpy(pad).to(tdev)
            outl = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outl, t)
            gathered = torch.cat(outl, dim=0).cpu().numpy()
        assert gathered.shape == (world * pad.shape[0], d)
        assert np.array_equal(gathered[rank * pad.shape[0]:rank * pad.shape[0] + B], u_local)

    if rank == 0:
        K = args.steps
        out = {
            "metric": "accepted ODE steps/sec (whole node), batched Brusselator N=32 nu=3 fp64",
            "value": accepted / elapsed,
            "unit": "accepted steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"Brusselator N=32 (d=64), nu=3, {'dense EKF1' if dense else 'isotropic EKF0'}, dynamic "
                                   f"calibration, filter, tol 1e-6, dt0 0.01, y0 phase family; global batch {total}"
                                   + (f", {B} trajectories on rank 0" if world > 1 else ""),
                       "batch_per_gpu": B, "global_batch": total, "parallelism": f"shard{world}",
                       "gather": gather_how if world > 1 else None,
                       "lq_kernel": ("one-launch (overlapping 64-row passes)" if one_launch_lq else "two-launch (32-row passes)") if dense else None,
                       "controller": "device-resident (controller kernel between the attempts, one host synchronisation for the "
                                     "K timed steps)" if device_ctl else "host (one round trip per step)",
                       "attempted_steps_per_s": total * K / elapsed,
                       "device_ms_per_step": dev_ms / K},
        }
        if dense:
            lq_ms = phase[2] / K
            flops = algorithmic_flops_lq(D, d) * B
            achieved = flops / (lq_ms * 1e-3) / 1e12
            # the north star also asks for the fraction of the HBM roofline: whole step (all dense kernels), algorithmic bytes
            # (state factor read + written once, 2 * 8 * D (D - d) per trajectory) and the bytes the PMC pass measured
            step_ms = dev_ms / K
            alg_bytes = 2.0 * 8.0 * D * (D - d) * B  # state factor Ls (D x (D - d)) read once and written once
            meas_bytes = measured_traffic("dense_", B)
            hbm = {"peak_GBps": HBM_PEAK_GBS, "streaming_ceiling_GBps": 4000.0,
                   "streaming_ceiling_source": "scripts/hbm_probe.hip: read+write in the strip layout, no compute",
                   "algorithmic_bytes_per_step": alg_bytes, "algorithmic_GBps": alg_bytes / (step_ms * 1e-3) / 1e9,
                   "algorithmic_frac": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "measured_bytes_per_step": meas_bytes}
            if meas_bytes:
                hbm["measured_GBps"] = meas_bytes / (step_ms * 1e-3) / 1e9
                hbm["measured_frac"] = hbm["measured_GBps"] / HBM_PEAK_GBS
            out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "hbm": hbm,
                               "peak_sustained_by_probe": {"value": 77.8, "unit": "TFLOP/s", "source": "scripts/mfma_probe.hip, profiles/r04_mfma_probe.txt: "
                                                           "v_mfma_f64_16x16x4_f64 with operands in registers, 1.7 s run, not re-measured in this run"},
                               "frac": achieved / FP64_PEAK_TFLOPS, "traffic": measured_traffic("dense_lq4_mfma_kernel" if one_launch_lq else "dense_lq2_mfma_kernel", B),
                               "traffic_source": "profiles/" + next((n for n in PMC_SUMMARIES if os.path.exists(os.path.join(ROOT, "profiles", n))), "-")
                                                 + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, separate passes; not re-measured in this run)",
                               "kernel": "dense_lq4_mfma_kernel<16,4> (one launch)" if one_launch_lq else
                                         "dense_lq2_mfma_kernel<16,true> + <16,false> (one LQ, two launches)",
                               "algorithmic_flops_per_launch": flops,
                               "algorithmic_flops_note": "2 D^2 (D - d) per trajectory: LQ on the D - d columns of the rank-deficient state "
                                                         "factor (r02 counted 2 D^3 for the LQ on D columns); time = mean over the K timed attempts",
                               "frac_r02_count": 2.0 * D**3 * B / (lq_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                               "phase_ms": {"begin": phase[0] / K, "lq_input": phase[1] / K, "lq": phase[2] / K,
                                            "correct": phase[3] / K}}
        else:
            n = NU + 1
            byts = 2 * 8 * (n * d + n * n) * B
            achieved = byts / (dev_ms / K * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "iso_step_kernel"}
        if world == 1 and not args.no_extras and args.diag_fixed_dt == 0.0:
            out.update(extras(args, batch, loop, active, dense, device_ctl, local))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
