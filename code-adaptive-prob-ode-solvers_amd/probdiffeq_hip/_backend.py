"""ctypes binding of the C-ABI in include/pnode.h (libpnode_hip.so, built by csrc/Makefile).

The product path has NO CPU fallback: if the shared library is missing, or no GPU is visible when a
batch is created, this module raises. (The CPU restatement under oracle/ is test infrastructure.)
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "libpnode_hip.so")

MODEL = {"isotropic": 0, "dense": 1}
CORRECTION = {"ts0": 0, "ts1": 1}
STRATEGY = {"filter": 0, "fixedpoint": 1}
CALIBRATION = {"none": 0, "dynamic": 1}

SYMBOLS = (
    "pnode_last_error pnode_version pnode_device_count pnode_create pnode_destroy pnode_set_initial "
    "pnode_set_tolerances pnode_attempt_step pnode_user_extrapolate pnode_user_set_vf pnode_commit pnode_interpolate pnode_checkpoint_at_step "
    "pnode_solve_fixed_grid pnode_solve_adaptive_save_at pnode_solve_adaptive_save_at_device pnode_read_mean pnode_read_time pnode_read_output_scale pnode_read_factor "
    "pnode_read_checkpoint_means pnode_read_checkpoint_factors pnode_smooth_checkpoint_means pnode_last_step_ms pnode_set_phase_timing "
    "pnode_last_phase_ms pnode_set_generic_path pnode_stream pnode_advance_device pnode_debug_set_lq_spin_limit pnode_phase_ms_mean pnode_sample_checkpoints_n pnode_set_smoother_pipeline pnode_read_checkpoint_transitions"
).split()


class PnodeControl(C.Structure):
    _fields_ = [
        ("safety", C.c_double),
        ("factor_min", C.c_double),
        ("factor_max", C.c_double),
        ("power_integral_unscaled", C.c_double),
        ("power_proportional_unscaled", C.c_double),
        ("clip", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PnodeConfig(C.Structure):
    _fields_ = [
        ("batch", C.c_int32),
        ("d", C.c_int32),
        ("nu", C.c_int32),
        ("ode_order", C.c_int32),
        ("model", C.c_int32),
        ("correction", C.c_int32),
        ("strategy", C.c_int32),
        ("calibration", C.c_int32),
        ("vf_id", C.c_int32),
        ("num_vf_params", C.c_int32),
        ("vf_params_per_trajectory", C.c_int32),
        ("num_checkpoints", C.c_int32),
        ("device", C.c_int32),
        ("reserved", C.c_int32),
        ("vf_params", C.POINTER(C.c_double)),
    ]


_lib = None

# A/B testing of the dense device paths through the high-level API: every dense Batch created while this is non-zero
# calls pnode_set_generic_path with it (1 = generic kernels, 2 = large-D kernels, 3 = MFMA kernels with the two-launch LQ).
# Tests set it with monkeypatch.setattr; nothing reads the environment.
dense_path = 0


def load_library():
    """Load libpnode_hip.so; raises OSError with a build hint if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise OSError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C code-adaptive-prob-ode-solvers_amd/csrc` (there is no CPU fallback)."
        )
    lib = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    u8p = C.POINTER(C.c_uint8)
    i32p = C.POINTER(C.c_int32)
    i64p = C.POINTER(C.c_int64)
    vp = C.c_void_p
    lib.pnode_last_error.restype = C.c_char_p
    lib.pnode_last_error.argtypes = []
    lib.pnode_version.restype = C.c_int
    lib.pnode_device_count.restype = C.c_int
    lib.pnode_create.argtypes = [C.POINTER(PnodeConfig), C.POINTER(vp)]
    lib.pnode_destroy.argtypes = [vp]
    lib.pnode_set_initial.argtypes = [vp, dp, dp, dp]
    lib.pnode_set_tolerances.argtypes = [vp, dp, dp]
    lib.pnode_attempt_step.argtypes = [vp, dp, u8p, dp]
    lib.pnode_user_extrapolate.argtypes = [vp, dp, u8p, dp]
    lib.pnode_user_set_vf.argtypes = [vp, dp, dp]
    lib.pnode_commit.argtypes = [vp, u8p]
    lib.pnode_interpolate.argtypes = [vp, dp, i32p]
    lib.pnode_checkpoint_at_step.argtypes = [vp, i32p]
    lib.pnode_solve_fixed_grid.argtypes = [vp, dp, C.c_int32, C.c_int32, dp]
    lib.pnode_solve_adaptive_save_at.argtypes = [vp, dp, C.c_int32, C.c_int32, dp, C.POINTER(PnodeControl), C.c_double,
                                                 C.c_int64, i64p, i64p, u8p]
    lib.pnode_solve_adaptive_save_at_device.argtypes = [vp, dp, C.c_int32, C.c_int32, dp, C.POINTER(PnodeControl), C.c_double,
                                                        C.c_int32, i64p, i64p, u8p]
    lib.pnode_read_mean.argtypes = [vp, C.c_int32, dp]
    lib.pnode_read_time.argtypes = [vp, dp, i64p]
    lib.pnode_read_output_scale.argtypes = [vp, C.c_int32, dp]
    lib.pnode_read_factor.argtypes = [vp, C.c_int32, dp]
    lib.pnode_read_checkpoint_means.argtypes = [vp, dp]
    lib.pnode_read_checkpoint_factors.argtypes = [vp, dp]
    lib.pnode_read_checkpoint_transitions.argtypes = [vp, dp, dp, dp]
    lib.pnode_smooth_checkpoint_means.argtypes = [vp, i32p, dp]
    lib.pnode_last_step_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pnode_set_phase_timing.argtypes = [vp, C.c_int32]
    lib.pnode_set_generic_path.argtypes = [vp, C.c_int32]
    lib.pnode_set_smoother_pipeline.argtypes = [vp, C.c_int32]
    lib.pnode_debug_set_lq_spin_limit.argtypes = [vp, C.c_int32]
    lib.pnode_last_phase_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pnode_phase_ms_mean.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
    lib.pnode_stream.argtypes = [vp]
    lib.pnode_offgrid_means.argtypes = [vp, i32p, dp, C.c_int32, dp, dp, dp, i32p, u8p, dp]
    lib.pnode_sample_checkpoints.argtypes = [vp, i32p, C.c_int32, dp, dp]
    lib.pnode_sample_checkpoints_n.argtypes = [vp, i32p, C.c_int32, C.c_int32, dp, dp]
    lib.pnode_advance_device.argtypes = [vp, dp, dp, dp, C.POINTER(PnodeControl), C.c_double, C.c_int32, i32p, i64p]
    lib.pnode_stream.restype = vp
    for name in SYMBOLS:
        if name not in ("pnode_last_error", "pnode_stream"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _check(rc):
    if rc >= 0:
        return rc
    msg = load_library().pnode_last_error().decode()
    if rc == -1:
        raise ValueError(msg)
    raise RuntimeError(f"pnode error {rc}: {msg}")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class Batch:
    """Owner of one `pnode_batch` handle: B independent IVPs resident on one GPU."""

    def __init__(self, *, batch, d, nu, ode_order, model, correction, strategy, calibration, vf_id,
                 vf_params=None, num_checkpoints=0, device=None, checkpoint_factors=False):
        lib = load_library()
        self._lib = lib
        self._h = None
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        prm = None
        per_traj = 0
        nprm = 0
        if vf_params is not None and np.size(vf_params) > 0:
            prm = np.ascontiguousarray(vf_params, dtype=np.float64)
            if prm.ndim == 2:
                if prm.shape[0] != batch:
                    raise ValueError("per-trajectory vf_params must have shape (batch, P)")
                per_traj, nprm = 1, prm.shape[1]
            else:
                prm = prm.reshape(-1)
                nprm = prm.size
        cfg = PnodeConfig(
            batch=batch, d=d, nu=nu, ode_order=ode_order, model=MODEL[model], correction=CORRECTION[correction],
            strategy=STRATEGY[strategy], calibration=CALIBRATION[calibration], vf_id=vf_id, num_vf_params=nprm,
            vf_params_per_trajectory=per_traj, num_checkpoints=num_checkpoints, device=device, reserved=1 if checkpoint_factors else 0,
            vf_params=_dp(prm) if prm is not None else None,
        )
        h = C.c_void_p()
        _check(lib.pnode_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.B, self.d, self.nu, self.n = batch, d, nu, nu + 1
        self.ode_order = ode_order
        self.D = self.n * d
        self.K = num_checkpoints
        self.model, self.strategy = model, strategy
        self.device = device
        if model == "dense" and dense_path:
            self.set_generic_path(dense_path)

    def close(self):
        if self._h is not None:
            self._lib.pnode_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state ---------------------------------------------------------------------------------
    def set_initial(self, tcoeffs, t0, output_scale):
        tc = _f64(tcoeffs, (self.B, self.n, self.d))
        _check(self._lib.pnode_set_initial(self._h, _dp(tc), _dp(_f64(t0, (self.B,))), _dp(_f64(output_scale, (self.B,)))))

    def set_tolerances(self, atol, rtol):
        a = _f64(np.broadcast_to(atol, (self.B,)))
        r = _f64(np.broadcast_to(rtol, (self.B,)))
        _check(self._lib.pnode_set_tolerances(self._h, _dp(a), _dp(r)))

    def _flags(self, a, name):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        if a.shape != (self.B,):
            raise ValueError(f"{name} must have shape ({self.B},), got {a.shape}")
        return a

    def attempt_step(self, dt, active, err_out):
        dt = _f64(dt, (self.B,))
        active = self._flags(active, "active")
        if not (isinstance(err_out, np.ndarray) and err_out.dtype == np.float64 and err_out.shape == (self.B,)
                and err_out.flags.c_contiguous and err_out.flags.writeable):
            raise ValueError(f"err_out must be a writeable C-contiguous float64 array of shape ({self.B},)")
        _check(self._lib.pnode_attempt_step(self._h, _dp(dt), active.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(err_out)))
        return err_out

    def user_extrapolate(self, dt, active, ode_order):
        """PNODE_VF_USER: extrapolated means m^-[:ode_order] (B, k, d) at which the host evaluates the vector field."""
        y = np.zeros((self.B, ode_order, self.d))
        active = self._flags(active, "active")
        _check(self._lib.pnode_user_extrapolate(self._h, _dp(_f64(dt, (self.B,))), active.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(y)))
        return y

    def user_set_vf(self, f, jac=None):
        f = _f64(f, (self.B, self.d))
        jp = None
        if jac is not None:
            jac = _f64(jac)
            if jac.size != self.B * self.d * self.ode_order * self.d:
                raise ValueError(f"jac must have {self.B} x {self.d} x {self.ode_order * self.d} entries, got shape {jac.shape}")
            jp = _dp(jac)
        _check(self._lib.pnode_user_set_vf(self._h, _dp(f), jp))

    def commit(self, accept):
        accept = self._flags(accept, "accept")
        _check(self._lib.pnode_commit(self._h, accept.ctypes.data_as(C.POINTER(C.c_uint8))))

    def _slots(self, slot):
        slot = np.ascontiguousarray(slot, dtype=np.int32)
        if slot.shape != (self.B,):
            raise ValueError(f"slot must have shape ({self.B},), got {slot.shape}")
        return slot

    def interpolate(self, t, slot):
        slot = self._slots(slot)
        _check(self._lib.pnode_interpolate(self._h, _dp(_f64(t, (self.B,))), slot.ctypes.data_as(C.POINTER(C.c_int32))))

    def checkpoint_at_step(self, slot):
        slot = self._slots(slot)
        _check(self._lib.pnode_checkpoint_at_step(self._h, slot.ctypes.data_as(C.POINTER(C.c_int32))))

    def solve_adaptive_save_at(self, save_at, dt0, control, numerical_zero, max_attempts=0, device_controller=False):
        """Native driver loop (controller + checkpoints on the host, in C++; or, opt-in, the controller on the
        device). Returns (num_steps, num_attempts, failed)."""
        save_at = _f64(save_at)
        per_traj = int(save_at.ndim == 2)
        if save_at.ndim not in (1, 2) or (per_traj and save_at.shape[0] != self.B):
            raise ValueError(f"save_at must have shape (K+1,) or ({self.B}, K+1), got {save_at.shape}")
        num_save = save_at.shape[-1]
        ctrl = PnodeControl(control.safety, control.factor_min, control.factor_max, control.power_integral_unscaled,
                            control.power_proportional_unscaled, int(control.clip), 0)
        steps = np.zeros((self.B, num_save), dtype=np.int64)
        attempts = np.zeros(self.B, dtype=np.int64)
        failed = np.zeros(self.B, dtype=np.uint8)
        dt0 = _f64(np.broadcast_to(dt0, (self.B,)))
        fn = self._lib.pnode_solve_adaptive_save_at_device if device_controller else self._lib.pnode_solve_adaptive_save_at
        _check(fn(self._h, _dp(save_at), num_save, per_traj, _dp(dt0), C.byref(ctrl), float(numerical_zero), int(max_attempts),
                  steps.ctypes.data_as(C.POINTER(C.c_int64)), attempts.ctypes.data_as(C.POINTER(C.c_int64)),
                  failed.ctypes.data_as(C.POINTER(C.c_uint8))))
        return steps, attempts, failed.astype(bool)

    def advance_device(self, t_stop, dt, log_err_prev, control, numerical_zero, rounds, status=None):
        """`rounds` attempts with the controller on the device and ONE host synchronisation (pnode_advance_device).
        dt and log_err_prev are updated in place; returns (status, attempts)."""
        if not (isinstance(dt, np.ndarray) and dt.dtype == np.float64 and dt.shape == (self.B,) and dt.flags.c_contiguous):
            raise ValueError(f"dt must be a C-contiguous float64 array of shape ({self.B},) (updated in place)")
        if not (isinstance(log_err_prev, np.ndarray) and log_err_prev.dtype == np.float64 and log_err_prev.shape == (self.B,)
                and log_err_prev.flags.c_contiguous):
            raise ValueError(f"log_err_prev must be a C-contiguous float64 array of shape ({self.B},) (updated in place)")
        ts = _f64(np.broadcast_to(t_stop, (self.B,)))
        st = np.zeros(self.B, dtype=np.int32) if status is None else np.ascontiguousarray(status, dtype=np.int32)
        if st.shape != (self.B,):
            raise ValueError(f"status must have shape ({self.B},)")
        att = np.zeros(self.B, dtype=np.int64)
        ctrl = PnodeControl(control.safety, control.factor_min, control.factor_max, control.power_integral_unscaled,
                            control.power_proportional_unscaled, int(control.clip), 0)
        _check(self._lib.pnode_advance_device(self._h, _dp(ts), _dp(dt), _dp(log_err_prev), C.byref(ctrl), float(numerical_zero),
                                              int(rounds), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                              att.ctypes.data_as(C.POINTER(C.c_int64))))
        return st, att

    def solve_fixed_grid(self, grid):
        grid = _f64(grid)
        per_traj = int(grid.ndim == 2)
        if per_traj and grid.shape[0] != self.B:
            raise ValueError("per-trajectory grid must have shape (batch, G)")
        G = grid.shape[-1]
        out = np.empty((self.B, G, self.d))
        _check(self._lib.pnode_solve_fixed_grid(self._h, _dp(grid), G, per_traj, _dp(out)))
        return out

    # --- read-back -----------------------------------------------------------------------------
    def read_mean(self, which=0):
        out = np.empty((self.B, self.n, self.d))
        _check(self._lib.pnode_read_mean(self._h, which, _dp(out)))
        return out

    def read_time(self):
        t = np.empty(self.B)
        s = np.empty(self.B, dtype=np.int64)
        _check(self._lib.pnode_read_time(self._h, _dp(t), s.ctypes.data_as(C.POINTER(C.c_int64))))
        return t, s

    def read_output_scale(self, which=0):
        out = np.empty(self.B)
        _check(self._lib.pnode_read_output_scale(self._h, which, _dp(out)))
        return out

    def read_factor(self, which=0):
        m = self.D if self.model == "dense" else self.n
        out = np.empty((self.B, m, m))
        _check(self._lib.pnode_read_factor(self._h, which, _dp(out)))
        return out

    def read_checkpoint_means(self):
        out = np.zeros((self.B, self.K, self.n, self.d))
        _check(self._lib.pnode_read_checkpoint_means(self._h, _dp(out)))
        return out

    def read_checkpoint_factors(self):
        """Upper factors R (B, K, m, m) with Sigma = R^T R at the checkpoints (m = n isotropic, m = n d dense filter)."""
        m = self.D if self.model == "dense" else self.n
        out = np.zeros((self.B, self.K, m, m))
        _check(self._lib.pnode_read_checkpoint_factors(self._h, _dp(out)))
        return out

    def read_checkpoint_transitions(self):
        """The stored backward transitions x_{c-1} | x_c ~ N(G x + xi, Lambda) (pnode_read_checkpoint_transitions).
        isotropic: G (B, K, n, n), xi (B, K, n, d), Lam (B, K, n, n) upper triangular, Lambda = Lam^T Lam (x) I_d;
        dense: G (B, K, D, D), xi (B, K, D), Lam (B, K, D, D) lower triangular, Lambda = Lam Lam^T."""
        m = self.D if self.model == "dense" else self.n
        G = np.zeros((self.B, self.K, m, m))
        xi = np.zeros((self.B, self.K, self.D) if self.model == "dense" else (self.B, self.K, self.n, self.d))
        Lam = np.zeros((self.B, self.K, m, m))
        _check(self._lib.pnode_read_checkpoint_transitions(self._h, _dp(G), _dp(xi), _dp(Lam)))
        return G, xi, Lam

    def smooth_checkpoint_means(self, num_filled):
        nf = self._slots(num_filled)
        out = np.zeros((self.B, self.K + 1, self.n, self.d))
        _check(self._lib.pnode_smooth_checkpoint_means(self._h, nf.ctypes.data_as(C.POINTER(C.c_int32)), _dp(out)))
        return out

    def offgrid_means(self, num_filled, init_means, t, t_lo, t_hi, idx, at_grid):
        """Off-grid marginal means (B, Q, n, d) of the stored posterior (pnode_offgrid_means)."""
        nf = self._slots(num_filled)
        t = _f64(t)
        if t.ndim != 2 or t.shape[0] != self.B:
            raise ValueError(f"t must have shape ({self.B}, Q)")
        Q = t.shape[1]
        t_lo, t_hi = _f64(t_lo, (self.B, Q)), _f64(t_hi, (self.B, Q))
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        ag = np.ascontiguousarray(at_grid, dtype=np.uint8)
        if idx.shape != (self.B, Q) or ag.shape != (self.B, Q):
            raise ValueError(f"idx and at_grid must have shape ({self.B}, {Q})")
        init = _f64(init_means, (self.B, self.n, self.d))
        out = np.zeros((self.B, Q, self.n, self.d))
        _check(self._lib.pnode_offgrid_means(self._h, nf.ctypes.data_as(C.POINTER(C.c_int32)), _dp(init), Q, _dp(t), _dp(t_lo), _dp(t_hi),
                                             idx.ctypes.data_as(C.POINTER(C.c_int32)), ag.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(out)))
        return out

    def sample_checkpoints(self, num_filled, eps):
        """Joint posterior samples (B, S, P, n, d) from base samples eps of the same shape; P >= max(num_filled) + 1 points per
        sample (pnode_sample_checkpoints_n; P = K + 1 is the store's capacity)."""
        nf = self._slots(num_filled)
        eps = _f64(eps)
        if eps.ndim != 5 or eps.shape[0] != self.B or eps.shape[3:] != (self.n, self.d) or eps.shape[2] < int(nf.max()) + 1:
            raise ValueError(f"eps must have shape ({self.B}, S, P >= {int(nf.max()) + 1}, {self.n}, {self.d})")
        out = np.zeros_like(eps)
        _check(self._lib.pnode_sample_checkpoints_n(self._h, nf.ctypes.data_as(C.POINTER(C.c_int32)), eps.shape[2], eps.shape[1],
                                                    _dp(eps), _dp(out)))
        return out

    def set_smoother_pipeline(self, mode):
        """0 wavefront per trajectory (default), 1 workgroup per trajectory (pipelined); include/pnode.h."""
        _check(self._lib.pnode_set_smoother_pipeline(self._h, int(mode)))

    def set_generic_path(self, enabled=True):
        _check(self._lib.pnode_set_generic_path(self._h, int(enabled)))

    def debug_set_lq_spin_limit(self, polls):
        _check(self._lib.pnode_debug_set_lq_spin_limit(self._h, int(polls)))

    def set_phase_timing(self, enabled=True):
        _check(self._lib.pnode_set_phase_timing(self._h, int(enabled)))

    def last_phase_ms(self):
        ms = (C.c_float * 4)()
        _check(self._lib.pnode_last_phase_ms(self._h, ms))
        return [float(x) for x in ms]

    def phase_ms_mean(self):
        """(ms[4], attempts): phase durations averaged over every attempt of the last advance_device call."""
        ms, cnt = (C.c_float * 4)(), C.c_int64()
        _check(self._lib.pnode_phase_ms_mean(self._h, ms, C.byref(cnt)))
        return [float(x) for x in ms], int(cnt.value)

    def last_step_ms(self):
        ms = C.c_float()
        _check(self._lib.pnode_last_step_ms(self._h, C.byref(ms)))
        return float(ms.value)


def device_count():
    return _check(load_library().pnode_device_count())
