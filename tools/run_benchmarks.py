"""Run the reference's benchmark scripts (batched) on the GPU and print one JSON summary per run.
Usage: python tools/run_benchmarks.py [random_linear|sweeps|rocket|grasp|quadruped|all] [batch] [results.npz | outdir/]
The optional .npz holds, per benchmark, the reference's result Dict entries (random_linear_problem.jl:188)
as arrays: "<name>/time" (ms per MPC step for the batch) and "<name>/iter" (steps x instances).  With a
directory instead, the three sweeps are written as horizon_comp.h5, state_dim_comp.h5, control_dim_comp.h5 in
the shape of the reference's *.jld2 result files (results_io.py: `results` + `Ns`, readable by
benchmarks/plotting.jl::comparison_plot after a five-line HDF5.jl loader)."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import benchmarks as Bm
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = {}
raw = {}
if which in ("all", "random_linear"):
    r = Bm.run_random_linear(batch=B); raw["random_linear n=12 m=4 N=50"] = r
    out["random_linear n=12 m=4 N=50"] = Bm.summarise(r)
sweeps = None
if which in ("all", "sweeps"):
    sweeps = Bm.run_sweeps(batch=min(B, 64))
    for name, pts in sweeps.items():
        for k, r in pts.items():
            out["%s %s" % (name, k)] = Bm.summarise(r); raw["%s %s" % (name, k)] = r
if which in ("all", "rocket"):
    r = Bm.run_rocket(batch=B); raw["rocket N_mpc=21"] = r
    out["rocket N_mpc=21"] = Bm.summarise(r)
if which in ("all", "grasp"):
    r = Bm.run_grasp(batch=min(B, 64), N_cold=101, tf=10.0); raw["grasp N_mpc=21"] = r
    out["grasp N_mpc=21"] = Bm.summarise(r)
if which in ("all", "quadruped"):
    for nm, lin in (("quadruped N=15 pyramids", True), ("quadruped N=15 cones", False)):
        r = Bm.run_quadruped(batch=B, linearized_friction=lin); raw[nm] = r
        out[nm] = Bm.summarise(r)
for k, v in out.items():
    print(json.dumps({"benchmark": k, **v}), flush=True)
if len(sys.argv) > 3 and sys.argv[3].endswith(os.sep) or (len(sys.argv) > 3 and os.path.isdir(sys.argv[3])):
    from altro_mpc_icra2021_amd import results_io
    os.makedirs(sys.argv[3], exist_ok=True)
    for name, fn in (("horizon", "horizon_comp.h5"), ("state_dim", "state_dim_comp.h5"), ("control_dim", "control_dim_comp.h5")):
        if sweeps:
            pts = sweeps[name]
            print("wrote", results_io.write_results(os.path.join(sys.argv[3], fn), list(pts.keys()), list(pts.values())))
elif len(sys.argv) > 3:
    import numpy as np
    np.savez_compressed(sys.argv[3], **{"%s/%s" % (k, f): np.asarray(r[f]) for k, r in raw.items() for f in ("time", "iter", "solve_succeeded")})
