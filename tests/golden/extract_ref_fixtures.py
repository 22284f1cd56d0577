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
"""
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


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are already committed")
    g = extract_grasp()
    print("grasp arrays:", [(a["hdf5_shape"], len(a["values"])) for a in g["arrays"]])
    s = extract_iter_stats()
    print(json.dumps(s, indent=1))
