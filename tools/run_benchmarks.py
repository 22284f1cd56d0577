"""Run the reference's benchmark scripts (batched) on the GPU and print one JSON summary per run.
Usage: python tools/run_benchmarks.py [random_linear|sweeps|rocket|grasp|quadruped|all] [batch]"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
from altro_mpc_icra2021_amd import benchmarks as Bm
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = {}
if which in ("all", "random_linear"):
    out["random_linear n=12 m=4 N=50"] = Bm.summarise(Bm.run_random_linear(batch=B))
if which in ("all", "sweeps"):
    for name, pts in Bm.run_sweeps(batch=min(B, 64)).items():
        for k, r in pts.items():
            out["%s %s" % (name, k)] = Bm.summarise(r)
if which in ("all", "rocket"):
    out["rocket N_mpc=21"] = Bm.summarise(Bm.run_rocket(batch=B))
if which in ("all", "grasp"):
    out["grasp N_mpc=21"] = Bm.summarise(Bm.run_grasp(batch=min(B, 64), N_cold=101, tf=10.0))
if which in ("all", "quadruped"):
    out["quadruped N=15 pyramids"] = Bm.summarise(Bm.run_quadruped(batch=B))
    out["quadruped N=15 cones"] = Bm.summarise(Bm.run_quadruped(batch=B, linearized_friction=False))
for k, v in out.items():
    print(json.dumps({"benchmark": k, **v}), flush=True)
