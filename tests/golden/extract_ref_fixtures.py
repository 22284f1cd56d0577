#!/usr/bin/env python3
"""Extract small known-answer fixtures from the DATA files the reference repo ships
(JLD2 = HDF5 with a 512-byte user block).  No h5py here, so this is a minimal HDF5
v2-object-header scanner (recipe: SURVEY.md Appendix C).  Reads /root/reference (build
container only) and writes JSON next to this script; the JSON is what the tests read.

Fixtures produced:
  grasp_ref_traj.json       arrays of benchmarks/grasp_optimization/grasp_ref_traj.jld2
                            (the only solver OUTPUT trajectory stored in the reference)
  ref_iteration_stats.json  ALTRO iteration-count / error statistics of the warm-started MPC
                            runs in horizon_comp.jld2, state_dim_comp.jld2, control_dim_comp.jld2
  ref_rocket_step.json      benchmarks/rocket_landing/rocket.jld2 (`res` of ONE conic MPC step:
                            iterations, costs, ALTRO-vs-conic-solver errors, times; `tols`) and the
                            error-vs-tolerance table the same script wrote to
                            figures/rocket_solver_tol.tikz (run_simple_rocket.jl:146-206,222)
  ref_grasp_mpc_stats.json  benchmarks/grasp_optimization/grasp_benchmark_data.jld2: per comparison
                            solver (ECOS, COSMO, Mosek) and horizon N_mpc in Ns, ALTRO's per-step
                            iteration counts, ALTRO-vs-solver errors and times (grasp_mpc.jl:96-108)
"""
import re
import json
import os
import struct
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def scan_datasets(path):
    """Return list of (offset, shape_hdf5, dtype_class, dtype_size, raw_bytes) for every v2
    object header that carries dataspace + datatype + layout messages."""
    buf = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        pos = buf.find(b"OHDR", pos)
        if pos < 0:
            break
        try:
            ver = buf[pos + 4]
            flags = buf[pos + 5]
            if ver != 2:
                pos += 4
                continue
            p = pos + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szlen = 1 << (flags & 3)
            chunk0 = int.from_bytes(buf[p:p + szlen], "little")
            p += szlen
            end = p + chunk0
            shape = None
            dt = None
            data = None
            while p + 4 <= end:
                mtype = buf[p]
                msize = struct.unpack_from("<H", buf, p + 1)[0]
                p += 4
                if flags & 4:
                    p += 2
                body = buf[p:p + msize]
                if mtype == 0x01 and len(body) >= 4:  # dataspace
                    v, rank, fl = body[0], body[1], body[2]
                    if v == 2:
                        off = 4
                        shape = [int.from_bytes(body[off + 8 * i:off + 8 * i + 8], "little") for i in range(rank)]
                elif mtype == 0x03 and len(body) >= 8:  # datatype
                    cls = body[0] & 0xF
                    size = struct.unpack_from("<I", body, 4)[0]
                    dt = (cls, size)
                elif mtype == 0x08 and len(body) >= 2:  # layout
                    v, lc = body[0], body[1]
                    if v in (3, 4):
                        if lc == 0:
                            sz = struct.unpack_from("<H", body, 2)[0]
                            data = body[4:4 + sz]
                        elif lc == 1:
                            addr, sz = struct.unpack_from("<QQ", body, 2)
                            if addr != 0xFFFFFFFFFFFFFFFF:
                                data = buf[addr:addr + sz]
                p += msize
            if shape is not None and dt is not None and data is not None:
                out.append((pos, shape, dt[0], dt[1], data))
        except Exception:
            pass
        pos += 4
    return out


def numeric_arrays(path):
    res = []
    for pos, shape, cls, size, data in scan_datasets(path):
        if cls == 1 and size == 8:
            a = np.frombuffer(data, dtype="<f8")
        elif cls == 0 and size == 8:
            a = np.frombuffer(data, dtype="<i8")
        else:
            continue
        n = int(np.prod(shape)) if shape else 1
        if a.size < n or n == 0:
            continue
        res.append((pos, shape, a[:n].copy()))
    return res


def extract_grasp():
    path = os.path.join(REF, "benchmarks/grasp_optimization/grasp_ref_traj.jld2")
    arrs = numeric_arrays(path)
    f8 = [(pos, sh, a) for pos, sh, a in arrs if a.dtype.kind == "f"]
    out = {"source": "benchmarks/grasp_optimization/grasp_ref_traj.jld2",
           "written_by": "benchmarks/grasp_optimization/old/altro_cold_solve.jl:102-117",
           "arrays": [{"offset": pos, "hdf5_shape": sh, "values": a.tolist()} for pos, sh, a in f8]}
    json.dump(out, open(os.path.join(HERE, "grasp_ref_traj.json"), "w"))
    return out


def extract_iter_stats():
    out = {}
    for name in ["horizon_comp.jld2", "state_dim_comp.jld2", "control_dim_comp.jld2"]:
        path = os.path.join(REF, name)
        arrs = numeric_arrays(path)
        entries = []
        for pos, sh, a in arrs:
            # result Dicts hold 100x2 arrays (HDF5 shape [2,100]); column 1 = ALTRO
            if sh == [2, 100]:
                col1, col2 = a[:100], a[100:200]
                entries.append({"offset": pos, "kind": "int" if a.dtype.kind == "i" else "float",
                                "altro": col1.tolist(), "other": col2.tolist()})
        out[name] = entries
    json.dump(out, open(os.path.join(HERE, "ref_mpc_arrays_raw.json"), "w"))
    # summarise the integer arrays (= :iter) only
    summ = {}
    for name, entries in out.items():
        its = [e for e in entries if e["kind"] == "int"]
        summ[name] = [{"altro_median": float(np.median(e["altro"])), "altro_mean": float(np.mean(e["altro"])),
                       "altro_max": int(np.max(e["altro"])), "altro_min": int(np.min(e["altro"])),
                       "other_median": float(np.median(e["other"]))} for e in its]
    json.dump({"source": "reference root *.jld2, Dict key :iter (random_linear_problem.jl:171-172,188)",
               "stats": summ}, open(os.path.join(HERE, "ref_iteration_stats.json"), "w"), indent=1)
    return summ


def extract_rocket():
    path = os.path.join(REF, "benchmarks/rocket_landing/rocket.jld2")
    arrs = numeric_arrays(path)
    by_shape = {}
    for pos, sh, a in arrs:
        by_shape.setdefault((tuple(sh), a.dtype.kind), []).append(a.tolist())
    # res = Dict(:time [1x2], :iter [1x2], :cost [1x3], :err_traj [1x3], :status) of run_Rocket_MPC(num_iters=1)
    # (simple_rocket.jl:206); HDF5 shapes are reversed, column 1 = ALTRO, column 2 = the conic solver
    out = {"source": "benchmarks/rocket_landing/rocket.jld2 (run_simple_rocket.jl:206)",
           "iter_altro_other": by_shape[((2, 1), "i")][0],
           "cost_altro_equiv_solver": by_shape[((3, 1), "f")][0],
           "err_traj_state_control_dynamics": by_shape[((3, 1), "f")][1],
           "time_ms_altro_other": by_shape[((2, 1), "f")][0],
           "tols": by_shape[((6,), "f")][0]}
    # tol_comp is stored as an SMatrix (a compound type the scanner skips); the same numbers were written as
    # plot coordinates by run_simple_rocket.jl:213-226
    tikz = open(os.path.join(REF, "figures/rocket_solver_tol.tikz")).read()
    legend = re.search(r"\\legend\{(.*)\}", tikz).group(1)
    names = re.findall(r"\{(\w+)\}", legend)
    blocks = re.findall(r"coordinates \{(.*?)\}", tikz, re.S)
    out["tol_comp"] = {"source": "figures/rocket_solver_tol.tikz",
                       "what": "max(state, control) inf-norm error of each solver's one-step MPC solution at solver tolerance tol"}
    for name, blk in zip(names, blocks):
        pts = re.findall(r"\(([^,]+),([^)]+)\)", blk)
        out["tol_comp"][name] = [[float(a), float(b)] for a, b in pts]
    json.dump(out, open(os.path.join(HERE, "ref_rocket_step.json"), "w"), indent=1)
    return out


def extract_grasp_benchmark():
    path = os.path.join(REF, "benchmarks/grasp_optimization/grasp_benchmark_data.jld2")
    arrs = numeric_arrays(path)
    Ns = [a.tolist() for pos, sh, a in arrs if sh == [5] and a.dtype.kind == "i"][0]
    # results[solver][N] = (Dict(:time [T x 2], :iter [T], :err_traj [T x 2]), ...), T = 251 - N_mpc
    # (grasp_benchmark.jl:66-85, grasp_mpc.jl:96-108); file order: :iter, :err_traj, :time per run
    runs, cur = [], None
    for pos, sh, a in arrs:
        if a.dtype.kind != "f" or not sh or sh[-1] < 100:
            continue
        if len(sh) == 1:
            cur = {"steps": sh[0], "iter": [int(v) for v in a]}
            runs.append(cur)
        elif cur is not None and sh == [2, cur["steps"]]:
            T = cur["steps"]
            if "err_state" not in cur:
                cur["err_state"], cur["err_control"] = a[:T].tolist(), a[T:2 * T].tolist()
            else:
                cur["time_ms_altro"], cur["time_ms_other"] = a[:T].tolist(), a[T:2 * T].tolist()
    assert len(runs) == 15 and all("time_ms_other" in r for r in runs)
    names = ["ECOS", "COSMO", "Mosek"]          # order of `optimizers`, grasp_benchmark.jl:36-61
    out = {"source": "benchmarks/grasp_optimization/grasp_benchmark_data.jld2 (grasp_benchmark.jl:66-88)", "Ns": Ns, "runs": []}
    for i, r in enumerate(runs):
        it = np.array(r["iter"])
        out["runs"].append({"solver": names[i // 5], "N_mpc": Ns[i % 5], "steps": r["steps"], "iter": r["iter"],
                            "iter_median": float(np.median(it)), "iter_mean": float(it.mean()), "iter_max": int(it.max()),
                            "iter_min": int(it.min()),
                            "err_state_median": float(np.median(r["err_state"])), "err_control_median": float(np.median(r["err_control"])),
                            "time_ms_altro_mean": float(np.mean(r["time_ms_altro"])), "time_ms_other_mean": float(np.mean(r["time_ms_other"]))})
    json.dump(out, open(os.path.join(HERE, "ref_grasp_mpc_stats.json"), "w"))
    return out


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are already committed")
    g = extract_grasp()
    print("grasp arrays:", [(a["hdf5_shape"], len(a["values"])) for a in g["arrays"]])
    s = extract_iter_stats()
    print(json.dumps(s, indent=1))
    r = extract_rocket()
    print(json.dumps(r, indent=1))
    gb = extract_grasp_benchmark()
    for run in gb["runs"]:
        print({k: v for k, v in run.items() if k != "iter"})
