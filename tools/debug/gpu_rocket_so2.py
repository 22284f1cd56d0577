"""configs[2] (rocket landing N_mpc = 100, batch 4096) with the two cone-Hessian variants: throughput, success, iterations"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
for so2 in (1, 0):
    altro.benchmarks.ROCKET_MPC_OPTS["soc_second_order"] = so2
    altro.benchmarks.ROCKET_COLD_OPTS["soc_second_order"] = so2
    d = bench.secondary_configs("rocket", 10, 3)[0]
    print("soc_second_order", so2, json.dumps({k: d[k] for k in ("value", "ms_per_step", "solve_succeeded_frac", "iterations_mean")}), "capped:", d.get("with_iteration_cap", {}).get("value"), flush=True)
