"""ctypes binding of libaltro_hip.so (C-ABI: include/altro_batch.h).

There is no fallback: if the HIP library has not been built, or no HIP device is usable, the
calls raise.  Nothing in this package computes a solve on the CPU.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("ALTRO_HIP_LIB") or os.path.join(CSRC, "libaltro_hip.so")

TRACE_LEN = 16

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_STATE, ERR_INTERNAL = range(6)
CON_BOX, CON_LINEAR, CON_SOC = 0, 1, 2
SENSE_EQ, SENSE_INEQ = 0, 1

STATUS_NAMES = ["UNSOLVED", "SOLVE_SUCCEEDED", "MAX_ITERATIONS", "MAX_ITERATIONS_OUTER",
                "MAXIMUM_COST", "STATE_LIMIT", "CONTROL_LIMIT", "NO_PROGRESS", "COST_INCREASE"]
SOLVE_SUCCEEDED = 1

EXPORTS = [
    "altro_default_opts", "altro_batch_create", "altro_batch_destroy", "altro_last_error",
    "altro_batch_set_dynamics", "altro_batch_set_tracking_cost", "altro_batch_add_constraint",
    "altro_batch_update_constraint_data", "altro_batch_set_initial_state",
    "altro_batch_set_reference", "altro_batch_set_initial_trajectory", "altro_batch_shift_fill",
    "altro_batch_set_options", "altro_batch_solve", "altro_batch_solve_async",
    "altro_batch_synchronize", "altro_batch_get_states", "altro_batch_get_controls",
    "altro_batch_get_duals", "altro_batch_set_duals", "altro_batch_get_stats",
    "altro_batch_get_alpha_trace", "altro_batch_get_gains", "altro_batch_last_solve_ms", "altro_batch_timing_reset", "altro_batch_timing_get",
    "altro_batch_get_work_counters", "altro_batch_get_wave_cycles", "altro_batch_get_solve_counters", "altro_mpc_run_async",
    "altro_mpc_set_noise_model", "altro_mpc_set_shift", "altro_mpc_set_track", "altro_mpc_set_noise",
    "altro_mpc_step_async", "altro_batch_get_initial_state", "altro_batch_get_stream",
    "altro_mpc_prepare_async", "altro_batch_benchmark_solve", "altro_mpc_set_dynamics_track",
    "altro_batch_get_confirm_counter", "altro_batch_get_reuse_counter", "altro_batch_get_polish_stats",
    "altro_debug_set", "altro_batch_get_polish_dual_residuals",
]
"""every symbol include/altro_batch.h declares"""


class Dims(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32)]


class Opts(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "cost_tolerance", "cost_tolerance_intermediate", "gradient_tolerance",
        "gradient_tolerance_intermediate", "constraint_tolerance", "penalty_initial",
        "penalty_scaling", "penalty_max", "dual_max", "line_search_lower_bound",
        "line_search_upper_bound", "max_cost_value", "max_state_value", "max_control_value",
        "bp_reg_initial", "bp_reg_increase_factor", "bp_reg_max", "bp_reg_min", "bp_reg_fp")] + \
        [(k, C.c_int32) for k in (
            "iterations", "iterations_inner", "iterations_outer", "iterations_linesearch",
            "dJ_counter_limit", "reset_duals", "reset_penalties", "bp_reg", "soc_second_order", "strict", "kickout_max_penalty",
            "projected_newton")] + \
        [(k, C.c_double) for k in ("projected_newton_tolerance", "active_set_tolerance_pn", "rho_chol", "rho_primal", "r_threshold")]


class AltroError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libaltro_hip error {code}: {msg}")
        self.code = code


def build(force=False, verbose=False):
    """Generate the DPP block include and compile the HIP library for gfx950, in tree.  The library is several translation
    units (altro_batch.hip: the C-ABI, the 16-lane kernels, the polish; wide_inst.hip once per group of one-wave-per-instance
    kernels, solve_wide.h ALTRO_WIDE_KERNELS) compiled side by side -- one after the other they take ~6 minutes."""
    srcs = [os.path.join(CSRC, f) for f in ("altro_batch.hip", "wide_inst.hip", "solve_dpp16.h", "solve_wide.h", "wide_backend.h", "launch_ring.h", "pn_polish.h", "pn_wide.h", "gen_dpp_blocks.py")]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "altro_batch.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["python3", os.path.join(CSRC, "gen_dpp_blocks.py"), os.path.join(CSRC, "dpp_blocks.inc")])
    # -amdgpu-mfma-vgpr-form: keep the FP64 MFMA accumulators of solve_wide.h in VGPRs; without it hipcc
    # (ROCm 7.2) shuttles them through AGPRs around every MFMA of the k loop (64 v_accvgpr moves per step)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(CSRC, "solve_wide.h")) as f:
        ntu = int(re.search(r"constexpr int kWideTUs = (\d+);", f.read()).group(1))
    jobs = [(os.path.join(objdir, "altro_batch.o"), ["hipcc"] + flags + ["-c", "-o", os.path.join(objdir, "altro_batch.o"), os.path.join(CSRC, "altro_batch.hip")])]
    for k in range(ntu):
        o = os.path.join(objdir, "wide_inst_%d.o" % k)
        jobs.append((o, ["hipcc"] + flags + ["-DALTRO_WIDE_TU=%d" % k, "-c", "-o", o, os.path.join(CSRC, "wide_inst.hip")]))
    procs = []
    for o, cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + [o for o, _ in jobs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """Load libaltro_hip.so.  Raises if it is missing: there is no CPU path to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AltroError(-1, f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    H = C.c_void_p
    L.altro_default_opts.argtypes = [C.POINTER(Opts)]
    L.altro_batch_create.argtypes = [C.POINTER(Dims), C.POINTER(Opts), C.c_int32, C.POINTER(H)]
    L.altro_batch_destroy.argtypes = [H]
    L.altro_last_error.argtypes = [H]
    L.altro_last_error.restype = C.c_char_p
    L.altro_batch_set_dynamics.argtypes = [H, dp, dp, dp, C.c_int32, C.c_int32]
    L.altro_batch_set_tracking_cost.argtypes = [H, dp, dp, dp, C.c_double]
    L.altro_batch_add_constraint.argtypes = [H, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             dp, dp, dp, dp, C.c_int32, ip]
    L.altro_batch_update_constraint_data.argtypes = [H, C.c_int32, dp, dp]
    L.altro_batch_set_initial_state.argtypes = [H, dp]
    L.altro_batch_get_initial_state.argtypes = [H, dp]
    L.altro_batch_set_reference.argtypes = [H, dp, dp]
    L.altro_batch_set_initial_trajectory.argtypes = [H, dp, dp]
    L.altro_batch_shift_fill.argtypes = [H, C.c_int32, C.c_int32]
    L.altro_batch_set_options.argtypes = [H, C.POINTER(Opts)]
    L.altro_batch_solve.argtypes = [H]
    L.altro_batch_solve_async.argtypes = [H]
    L.altro_batch_synchronize.argtypes = [H]
    L.altro_batch_get_states.argtypes = [H, dp]
    L.altro_batch_get_controls.argtypes = [H, dp]
    L.altro_batch_get_duals.argtypes = [H, C.c_int32, dp]
    L.altro_batch_set_duals.argtypes = [H, C.c_int32, dp]
    L.altro_batch_get_stats.argtypes = [H, ip, ip, ip, dp, dp, dp, dp]
    L.altro_batch_get_alpha_trace.argtypes = [H, dp]
    L.altro_batch_get_gains.argtypes = [H, dp, dp]
    L.altro_batch_last_solve_ms.argtypes = [H, C.POINTER(C.c_float)]
    L.altro_batch_timing_reset.argtypes = [H]
    L.altro_batch_timing_get.argtypes = [H, C.POINTER(C.c_float), C.c_int32, ip]
    L.altro_batch_get_work_counters.argtypes = [H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.altro_batch_get_wave_cycles.argtypes = [H, C.POINTER(C.c_int64), C.c_int32, ip]
    i64 = C.POINTER(C.c_int64)
    L.altro_batch_get_solve_counters.argtypes = [H, i64, i64, i64]
    L.altro_mpc_run_async.argtypes = [H, C.c_int32, C.c_int32]
    L.altro_mpc_set_noise_model.argtypes = [H, C.c_int32, dp, ip]
    L.altro_mpc_set_shift.argtypes = [H, C.c_int32]
    L.altro_mpc_set_track.argtypes = [H, dp, dp, C.c_int32]
    L.altro_mpc_set_noise.argtypes = [H, dp, C.c_int32]
    L.altro_mpc_step_async.argtypes = [H, C.c_int32]
    L.altro_batch_get_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    L.altro_mpc_prepare_async.argtypes = [H, C.c_int32]
    L.altro_batch_get_confirm_counter.argtypes = [H, C.POINTER(C.c_int64)]
    L.altro_batch_get_reuse_counter.argtypes = [H, C.POINTER(C.c_int64)]
    L.altro_batch_get_polish_stats.argtypes = [H, ip, ip, dp]
    L.altro_mpc_set_dynamics_track.argtypes = [H, dp, dp, dp, C.c_int32, C.c_int32, C.c_int32]
    L.altro_batch_benchmark_solve.argtypes = [H, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    L.altro_debug_set.argtypes = [H, C.c_char_p, C.c_int32]
    L.altro_batch_get_polish_dual_residuals.argtypes = [H, dp, dp, ip]
    for name in EXPORTS:
        if name != "altro_last_error":
            getattr(L, name).restype = C.c_int32
    _lib = L
    return L


# Environment variables of the tests and measuring tools -> altro_debug_set keys.  The LIBRARY reads no environment; this
# harness forwards these variables to its explicit entry point whenever a solver is created (api.ALTROSolver), so that
# `ALTRO_NO_LONE=1 python tools/...` and pytest's monkeypatch.setenv keep working.  (variable, key, default, negate)
DEBUG_ENV = [
    ("ALTRO_NO_LONE", "no_lone", 0), ("ALTRO_NO_SHADOW", "no_shadow", 0), ("ALTRO_NO_RESYNC", "no_resync", 0),
    ("ALTRO_NO_GROUP", "no_group", 0), ("ALTRO_NO_REUSE", "no_reuse", 0), ("ALTRO_NO_QZ_PASS", "no_qz_pass", 0), ("ALTRO_NO_MATE_RANK", "no_mate_rank", 0), ("ALTRO_GROUP_MAX_STEPS", "group_max_steps", 32),
    ("ALTRO_DEBUG_TRACE_WAVE", "trace_wave", -1), ("ALTRO_FORCE_WIDE", "force_wide", 0),
    ("ALTRO_WIDE_COMPACT", "wide_compact", -1), ("ALTRO_WIDE_COOP", "wide_coop", -1),
    ("ALTRO_WIDE_STATIC_MASK", "wide_static_mask", -1), ("ALTRO_DEBUG_KEEP_GAINS", "keep_gains", 0),
]


def debug_set(key, value, handle=None):
    """altro_debug_set(handle or NULL, key, value); raises AltroError on a refusal"""
    L = lib()
    rc = L.altro_debug_set(handle, key.encode(), int(value))
    if rc:
        raise AltroError(rc, (L.altro_last_error(handle) or b"").decode())


def sync_debug_env():
    """forward the ALTRO_* diagnostic variables of the environment to altro_debug_set (defaults where unset)"""
    for var, key, default in DEBUG_ENV:
        v = os.environ.get(var)
        try:
            debug_set(key, int(v) if v not in (None, "") else default)
        except AltroError as e:
            # an older build of the library (ALTRO_HIP_LIB, the A/B tools) may not know a newer switch: fine while it is not asked for
            if e.code != ERR_INVALID_ARG or v not in (None, ""):
                raise
    gm = os.environ.get("ALTRO_GROUP_MODE")     # after no_group: a slot order switches grouping on
    if gm not in (None, ""):
        debug_set("group_mode", int(gm))
