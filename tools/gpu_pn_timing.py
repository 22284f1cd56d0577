"""Timing line of the projected-Newton polish (csrc/pn_polish.h, csrc/pn_wide.h): the grasp cold solve at the reference's
N = 251 (grasp_benchmark.jl:19-25), AL stage to a loose 1e-2, polish to 1e-6 -- device time of the solve with and without
the polish, on the 16-lane backend and forced onto the one-wave-per-instance backend.  python tools/gpu_pn_timing.py [batch]"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from helpers import rocket_gpu_problem
P = altro.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gp = P.gen_grasp_problem(N=251, tf=25.0)
rng = np.random.default_rng(5)
x0 = np.tile(gp.x0, (B, 1))
x0[:, 1:3] += 0.1 * rng.standard_normal((B, 2))
base = dict(cost_tolerance_intermediate=1e-5, penalty_initial=1.0, penalty_scaling=10.0, iterations=5000, iterations_outer=60)
out = {}
for backend, wide in (("16-lane (pn_polish.h)", 0), ("one wave per instance (pn_wide.h)", 1)):
    altro.debug_set("force_wide", wide)
    os.environ["ALTRO_FORCE_WIDE"] = str(wide)
    ms = {}
    for tag, opts in (("al_only", dict(base, constraint_tolerance=1e-2)),
                      ("al_plus_polish", dict(base, constraint_tolerance=1e-6, projected_newton=1, projected_newton_tolerance=1e-2))):
        sv = altro.ALTROSolver(rocket_gpu_problem(altro, gp, x0), altro.SolverOptions(**opts))
        altro.solve(sv)            # warm the clocks / code
        sv.close()
        sv = altro.ALTROSolver(rocket_gpu_problem(altro, gp, x0), altro.SolverOptions(**opts))
        altro.solve(sv)
        st = altro.stats(sv)
        ms[tag] = float(st.tsolve_ms)
        if tag == "al_plus_polish":
            ran, failed, res = altro.polish_stats(sv)
            d0, d1, df = altro.polish_dual_residuals(sv)
            ms["polished"] = int(ran.sum()); ms["failed"] = int(failed.sum()); ms["residual_max"] = float(res.max())
            ms["c_max"] = float(st.c_max.max()); ms["dual_residual_before_after"] = [float(d0.mean()), float(d1.mean())]
        sv.close()
    ms["polish_ms_per_instance"] = (ms["al_plus_polish"] - ms["al_only"]) / max(1, ms["polished"])
    out[backend] = ms
print(json.dumps({"workload": "grasp cold solve N = 251, batch %d; AL stage to 1e-2, polish to 1e-6" % B, "timing": out}))
