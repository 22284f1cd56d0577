#!/usr/bin/env python3
"""Emit dpp_blocks.inc: the cross-lane FP64 product blocks of the 16-lane-per-instance
AL-iLQR kernels, one `Blk<NX,NU>` specialisation per instantiated problem size.

Why generated text and not C++ templates: every block is ONE `asm volatile` statement whose
body is a fixed sequence of `v_fmac_f64_dpp ... row_newbcast:k` instructions.  hipcc
(ROCm 7.2) does not fold `v_mov_b64_dpp` into the consuming FMA (measured: the mov+fma form
runs at half the FP64 rate, the fused form at the full rate, tools/probes/), and an asm body
must be a string literal.  Keeping a block in one statement also fixes the instruction
order, which is what makes the CDNA hazard rules checkable by construction:

  * "VALU writes VGPR -> DPP reads that VGPR: 2 wait states" and "VALU writes EXEC -> DPP:
    5 wait states": each block opens with `s_nop 4`; no instruction inside a block writes a
    register that a later DPP operand of the block reads within two instructions (the DPP
    source operands are block inputs, never block outputs);
  * accumulators are visited round-robin, so the same accumulator is touched at most once
    every NACC >= 3 instructions (FP64 FMA dependent-issue distance).

All blocks must execute with EXEC = all ones (a DPP read from a disabled lane returns 0): the
kernels call them only from wave-uniform control flow.

Lane layout (per 16-lane DPP row = one MPC instance): lane j < NX owns state column j, lane
NX+a owns control column a, lanes >= NX+NU idle (carry zeros).
"""
import sys

SIZES = [(12, 4), (6, 3), (6, 6), (8, 4), (12, 3)]

DPP = "row_newbcast:{k} row_mask:0xf bank_mask:0xf"


def fmac(acc, src_dpp, src, k):
    return f"v_fmac_f64_dpp %[{acc}], %[{src_dpp}], %[{src}] " + DPP.format(k=k)


def emit_block(name, sig, body, outs, ins):
    """sig: C++ parameter list; outs/ins: lists of (asmname, c_expr)."""
    o = ", ".join(f'[{a}] "+v"({c})' for a, c in outs)
    i = ", ".join(f'[{a}] "v"({c})' for a, c in ins)
    lines = ['"s_nop 4\\n\\t"'] + [f'"{b}\\n\\t"' for b in body]
    s = f"  static __device__ __forceinline__ void {name}({sig}) {{\n    asm volatile(\n      "
    s += "\n      ".join(lines)
    s += f"\n      : {o}\n      : {i});\n  }}\n"
    return s


def gen(NX, NU):
    NZ = NX + NU
    assert NZ <= 16
    out = [f"template <> struct Blk<{NX}, {NU}> {{\n"]

    # ---- SG: w[i] += sum_k bcast_k(Sx[i]) * g[k],  i in 0..NX (row NX of Sx is the vector s)
    #      => w[0..NX-1] = column of S*G, w[NX] = (G' s)[lane]
    body = [fmac(f"w{i}", f"s{i}", f"g{k}", k) for k in range(NX) for i in range(NX + 1)]
    out.append(emit_block(
        "SG", f"double (&w)[{NX + 1}], const double (&Sx)[{NX + 1}], const double (&g)[{NX}]", body,
        [(f"w{i}", f"w[{i}]") for i in range(NX + 1)],
        [(f"s{i}", f"Sx[{i}]") for i in range(NX + 1)] + [(f"g{k}", f"g[{k}]") for k in range(NX)]))

    # ---- GtW: h[i] += sum_k bcast_i(g[k]) * w[k],  i in 0..NZ  (H = G' W, column per lane)
    body = [fmac(f"h{i}", f"g{k}", f"w{k}", i) for k in range(NX) for i in range(NZ)]
    out.append(emit_block(
        "GtW", f"double (&h)[{NZ}], const double (&g)[{NX}], const double (&w)[{NX + 1}]", body,
        [(f"h{i}", f"h[{i}]") for i in range(NZ)],
        [(f"g{k}", f"g[{k}]") for k in range(NX)] + [(f"w{k}", f"w[{k}]") for k in range(NX)]))

    # ---- CTG: h[i] += sum_a bcast_i(kd[a]) * T[a] + bcast_i(r[a]) * kd[a],  i in 0..NX
    #      (S = Qxx + K'(Quu K + Qux) + Qux' K, column per lane)
    body = []
    for a in range(NU):
        body += [fmac(f"h{i}", f"k{a}", f"t{a}", i) for i in range(NX)]
        body += [fmac(f"h{i}", f"r{a}", f"k{a}", i) for i in range(NX)]
    out.append(emit_block(
        "CTG", f"double (&h)[{NZ}], const double (&kd)[{NU}], const double (&T)[{NU}], const double (&r)[{NU}]",
        body, [(f"h{i}", f"h[{i}]") for i in range(NX)],
        [(f"k{a}", f"kd[{a}]") for a in range(NU)] + [(f"t{a}", f"T[{a}]") for a in range(NU)] +
        [(f"r{a}", f"r[{a}]") for a in range(NU)]))

    # ---- CTG0: h[i] += sum_a bcast_i(r[a]) * kd[a]   (the rho == 0 case: S = Qxx + Qux' K)
    body = []
    for a in range(NU):
        body += [fmac(f"h{i}", f"r{a}", f"k{a}", i) for i in range(NX)]
    out.append(emit_block(
        "CTG0", f"double (&h)[{NZ}], const double (&kd)[{NU}], const double (&r)[{NU}]",
        body, [(f"h{i}", f"h[{i}]") for i in range(NX)],
        [(f"k{a}", f"kd[{a}]") for a in range(NU)] + [(f"r{a}", f"r[{a}]") for a in range(NU)]))

    # ---- KDXT: acc[a][p] += sum_{j in part p} bcast_j(prod[a]) * one,  a < NU, j < NX, 3 parts
    #      prod[a] (x lane j) = K[a][j] * dx[j]  =>  sum_p acc[a][p] = (K dx)[a] on every lane
    NP = 3
    body = [fmac(f"c{a}_{j % NP}", f"p{a}", "one", j) for j in range(NX) for a in range(NU)]
    out.append(emit_block(
        "KDXT", f"double (&acc)[{NU}][{NP}], const double (&prod)[{NU}], const double& one", body,
        [(f"c{a}_{p}", f"acc[{a}][{p}]") for a in range(NU) for p in range(NP)],
        [(f"p{a}", f"prod[{a}]") for a in range(NU)] + [("one", "one")]))

    # ---- GZ: xn_p += sum_{j in part p} bcast_j(z) * grow[j]   (4 partial sums, j < NZ)
    NP = 4
    body = [fmac(f"a{j % NP}", "z", f"g{j}", j) for j in range(NZ)]
    out.append(emit_block(
        "GZ", f"double (&acc)[{NP}], const double& z, const double (&grow)[{NZ}]", body,
        [(f"a{p}", f"acc[{p}]") for p in range(NP)],
        [("z", "z")] + [(f"g{j}", f"grow[{j}]") for j in range(NZ)]))

    # ---- GTS: acc_p += sum_{k in part p} bcast_k(s) * g[k]   ((G' s)[lane]: the costate recursion of the
    #      gradient-only convergence check, 4 partial sums, k < NX)
    body = [fmac(f"a{k % NP}", "s", f"g{k}", k) for k in range(NX)]
    out.append(emit_block(
        "GTS", f"double (&acc)[{NP}], const double& s, const double (&g)[{NX}]", body,
        [(f"a{p}", f"acc[{p}]") for p in range(NP)],
        [("s", "s")] + [(f"g{k}", f"g[{k}]") for k in range(NX)]))

    # ---- HC: h[i] += sum_r bcast_i(acol[r]) * y[r],  i < NZ, r < 16   (H += A' (M A), column per lane)
    body = [fmac(f"h{i}", f"c{r}", f"y{r}", i) for r in range(16) for i in range(NZ)]
    out.append(emit_block(
        "HC", f"double (&h)[{NZ}], const double (&acol)[16], const double (&y)[16]", body,
        [(f"h{i}", f"h[{i}]") for i in range(NZ)],
        [(f"c{r}", f"acol[{r}]") for r in range(16)] + [(f"y{r}", f"y[{r}]") for r in range(16)]))


    # ---- lone-row blocks: ONE instance spread over the wave's four DPP rows (solve_dpp16.h backward_lone).
    # Row r of the wave owns rows i = r*RL + t (t < RL) of S, W and Qxx, row 0 also the vector s (slot RL), and the
    # rows NX + r*RQ + u (u < RQ) of [Qux Quu]; every row holds all 16 columns (lane j = column j as before).
    # Every output element is the same chain of FMAs in the same order as in SG / GtW / CTG0, so the lone pass
    # is bit-identical to the four-row pass.
    RL = (NX + 3) // 4
    RQ = (NU + 3) // 4
    out.append(f"  static constexpr int RL = {RL}, RQ = {RQ};\n")
    # SGL: w[t] += sum_k bcast_k(Sl[t]) * g[k],  t <= RL
    body = [fmac(f"w{t}", f"s{t}", f"g{k}", k) for k in range(NX) for t in range(RL + 1)]
    out.append(emit_block(
        "SGL", f"double (&w)[{RL + 1}], const double (&Sl)[{RL + 1}], const double (&g)[{NX}]", body,
        [(f"w{t}", f"w[{t}]") for t in range(RL + 1)],
        [(f"s{t}", f"Sl[{t}]") for t in range(RL + 1)] + [(f"g{k}", f"g[{k}]") for k in range(NX)]))
    # GTWL: h[t] += sum_k bcast_t(gp[k]) * wa[k],  t < RL + RQ  (gp: the row's own columns of G, permuted to lanes 0..)
    body = [fmac(f"h{t}", f"g{k}", f"w{k}", t) for k in range(NX) for t in range(RL + RQ)]
    out.append(emit_block(
        "GTWL", f"double (&h)[{RL + RQ}], const double (&gp)[{NX}], const double (&wa)[{NX + 1}]", body,
        [(f"h{t}", f"h[{t}]") for t in range(RL + RQ)],
        [(f"g{k}", f"gp[{k}]") for k in range(NX)] + [(f"w{k}", f"wa[{k}]") for k in range(NX)]))
    # CTGL0: h[t] += sum_a bcast_t(rp[a]) * kd[a],  t < RL  (rp: Qux rows permuted like gp)
    body = []
    for a in range(NU):
        body += [fmac(f"h{t}", f"r{a}", f"k{a}", t) for t in range(RL)]
    out.append(emit_block(
        "CTGL0", f"double (&h)[{RL + RQ}], const double (&kd)[{NU}], const double (&rp)[{NU}]",
        body, [(f"h{t}", f"h[{t}]") for t in range(RL)],
        [(f"k{a}", f"kd[{a}]") for a in range(NU)] + [(f"r{a}", f"rp[{a}]") for a in range(NU)]))

    out.append("};\n\n")
    return "".join(out)


def gen_common():
    # ROWSUM over all 16 lanes of a DPP row: acc_p += bcast_j(v) * one
    NP = 4
    body = [fmac(f"a{j % NP}", "v", "one", j) for j in range(16)]
    s = "struct BlkCommon {\n"
    s += emit_block("ROWSUM", f"double (&acc)[{NP}], const double& v, const double& one", body,
                    [(f"a{p}", f"acc[{p}]") for p in range(NP)], [("v", "v"), ("one", "one")])
    # ---- generic affine constraints: 16 constraint rows per knot, row r on lane r, 4 quads of 4
    # RS16: acc_p += sum_{r in part p} bcast_r(v) * coef[r]      ((A' g)[lane] with coef = column of A)
    body = [fmac(f"a{r % NP}", "v", f"c{r}", r) for r in range(16)]
    s += emit_block("RS16", f"double (&acc)[{NP}], const double& v, const double (&coef)[16]", body,
                    [(f"a{p}", f"acc[{p}]") for p in range(NP)], [("v", "v")] + [(f"c{r}", f"coef[{r}]") for r in range(16)])
    # YM: y[r] += sum_q bcast_r(m[q]) * acol[4*(r/4)+q]          (Y = M A, block-diagonal 4x4 M, row r of M on lane r)
    body = [fmac(f"y{r}", f"m{q}", f"c{4 * (r // 4) + q}", r) for q in range(4) for r in range(16)]
    s += emit_block("YM", "double (&y)[16], const double (&m)[4], const double (&acol)[16]", body,
                    [(f"y{r}", f"y[{r}]") for r in range(16)],
                    [(f"m{q}", f"m[{q}]") for q in range(4)] + [(f"c{r}", f"acol[{r}]") for r in range(16)])
    s += "};\n\n"
    return s


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "dpp_blocks.inc"
    txt = "// GENERATED by gen_dpp_blocks.py -- do not edit; edit the generator.\n"
    txt += "// FP64 DPP (row_newbcast) product blocks, one asm statement each.  See the generator's\n"
    txt += "// docstring for the hazard and EXEC rules these blocks rely on.\n\n"
    txt += "template <int NX, int NU> struct Blk;\n\n"
    txt += gen_common()
    for nx, nu in SIZES:
        txt += gen(nx, nu)
    open(path, "w").write(txt)


if __name__ == "__main__":
    main()
