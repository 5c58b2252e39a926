#!/usr/bin/env python3
"""Why does the CPU baseline of bench.py scale so badly on the GPU box (r02: 1.6x on 16 threads)? Runs the C oracle's adaptive
loop (oracle/pn_oracle_c.c) for 1, 2, 4, 8, 16 threads with both back ends, unpinned and pinned to distinct cores, and prints the
cgroup's throttling counters around every run.   python scripts/cpu_scaling_probe.py [seconds]"""
import ctypes as C, glob, os, sys
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("code-adaptive-prob-ode-solvers_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
import numpy as np
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
vf, tcs = bench.problem_batch(16, 0, 1024)
tc = np.ascontiguousarray(np.stack(tcs, axis=1))
lib = C.CDLL(os.path.join(ROOT, "oracle", "libpn_oracle_c.so"))
dp = C.POINTER(C.c_double)
lib.pnoc_bench_brusselator.restype = C.c_double
lib.pnoc_bench_brusselator.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
lib.pnoc_use_lapack.argtypes = [C.c_char_p]


def throttled():
    try:
        return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
    except OSError:
        return {}


def run(nt):
    acc, att = C.c_long(), C.c_long()
    t0 = throttled()
    el = lib.pnoc_bench_brusselator(32, 3, nt, tc.ctypes.data_as(dp), 1e-6, 0.01, secs, nt, C.byref(acc), C.byref(att))
    t1 = throttled()
    thr = int(t1.get("nr_throttled", 0)) - int(t0.get("nr_throttled", 0))
    thr_us = int(t1.get("throttled_usec", 0)) - int(t0.get("throttled_usec", 0))
    return acc.value / el, thr, thr_us / 1e6


print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None,
      "loadavg", open("/proc/loadavg").read().strip())
allowed = sorted(os.sched_getaffinity(0))
for backend in ("plain", "lapack"):
    if backend == "lapack":
        import scipy
        hits = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        assert hits and lib.pnoc_use_lapack(hits[0].encode()) == 1
    for pin in (False, True):
        os.sched_setaffinity(0, set(allowed[:: max(1, len(allowed) // 16)][:16]) if pin else set(allowed))
        row = []
        for nt in (1, 2, 4, 8, 16):
            v, thr, ts = run(nt)
            row.append(f"{nt}: {v:7.1f} ({v / nt:6.1f}/thread, throttled {thr}x {ts:.2f}s)")
        print(f"{backend:6s} pinned={pin}: " + " | ".join(row), flush=True)
os.sched_setaffinity(0, set(allowed))
