#!/usr/bin/env python3
"""Generates gpy_amd/csrc/potf2_asm.h: the 16x16 Cholesky + inverse of the factor in the registers of ONE wave
(the dependent chain under every 128-column step of dpotrf, GPy/util/linalg.py:58) as one hand-scheduled gfx950
instruction stream.

Why generated assembly: the chain is issue-bound on one wave (one fp64 VALU instruction per 4 cycles), and the compiler
  * never fuses the DPP row broadcast into its consumer: `v_mov_b64_dpp` + `v_fma_f64` per rank-1 element where ONE
    `v_fmac_f64_dpp` (VOP2, DP-ALU DPP, row_newbcast) does both, and `v_mov_b64_dpp` + readlanes + `v_rsq_f64` where
    `v_rsq_f64_dpp` takes the broadcast pivot directly;
  * carries the library rsqrt's class tests and two 64-bit selects per column that the algorithm does not need.
~1000 -> ~420 VALU instructions per 16 x 16 tile.

Data layout (unchanged from the C++ version it replaces): lane l works on row i = l & 15; a[c] = A[i][c] is mirrored in
the four 16-lane rows of the wave; the running right-hand side I -> L^-1 is split over them (lane group g = l >> 4 keeps
columns 4k + g in xs[k]).  Right-looking; column j + 1's pivot chain (rsqrt + one cubic refinement step) is software-
pipelined underneath the remaining rank-1 updates of column j.

Hazards the hardware does not interlock and the assembler does not see (inline asm is opaque to the hazard recogniser),
enforced by the tracker below: a DPP read of a VGPR written by a VALU instruction needs 2 wait states; the result of a
transcendental (v_rsq_f64) needs 1 wait state before a non-transcendental VALU reads it.

Upper triangle: rows i < c of a[c] are never read by another lane and come out as garbage (the caller never stores
them).  No pivot test inside the chain: a pivot that is not positive (or NaN) turns into NaN -- rsq of a negative number is
NaN, rsq(0) * 0 is NaN -- and poisons every later one, so the caller reads the index of the first bad pivot off the
diagonal of the result (the first L[j][j] that is not > 0).

    python tools/gen_potf2.py > gpy_amd/csrc/potf2_asm.h
"""
import argparse
import sys

BASE = 64          # first VGPR of the temporaries (fixed registers, listed as clobbers)
SB = 92            # first SGPR of the scalar temporaries


class Gen:
    def __init__(self):
        self.lines = []
        self.idx = 0                      # wait states issued so far
        self.wr = {}                      # reg name -> (idx of the VALU write, is_trans)
        self.nvalu = 0
        self.pad = 0                      # diagnostic: wait states forced in front of EVERY VALU instruction

    def _need(self, reads, dpp_reads):
        need = 0
        for r in dpp_reads:
            if r in self.wr:
                d = self.idx - self.wr[r][0]          # 1 = back to back
                need = max(need, 3 - d)
        for r in list(reads) + list(dpp_reads):
            if r in self.wr and self.wr[r][1]:
                d = self.idx - self.wr[r][0]
                need = max(need, 2 - d)
        return need

    def nop(self, n):
        if n <= 0:
            return
        self.lines.append("s_nop %d" % (n - 1))
        self.idx += n

    def valu(self, text, writes=(), reads=(), dpp_reads=(), trans=False):
        self.nop(max(self._need(reads, dpp_reads), self.pad))
        self.lines.append(text)
        for w in writes:
            self.wr[w] = (self.idx, trans)
        self.idx += 1
        self.nvalu += 1

    def salu(self, text):
        self.lines.append(text)
        self.idx += 1


def vp(n):
    return "v[%d:%d]" % (n, n + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="potf2_inv_16_asm")
    ap.add_argument("--pad", type=int, default=0, help="diagnostic: wait states in front of every VALU instruction")
    ap.add_argument("--dpp-rsq", action="store_true", help="v_rsq_f64_dpp instead of v_mov_b64_dpp + v_rsq_f64: assembles, but gfx950 "
                    "returns garbage for it (tools/hwprobe/potf2_probe.hip); diagnostic only")
    ap.add_argument("--plain-fmac", action="store_true", help="v_mov_b64_dpp + v_fmac_f64 instead of v_fmac_f64_dpp")
    args = ap.parse_args()
    g = Gen()
    g.pad = args.pad
    A = ["%%%d" % c for c in range(16)]           # operands 0..15: a[c]
    X = ["%%%d" % (16 + k) for k in range(4)]     # operands 16..19: xs[k]
    # temporaries, two sets (column parity): piv, y, t, e, u, rd, nl
    def T(p, k):
        return BASE + 16 * p + 2 * k
    PIV, Y, TT, E, U, RD, NL, NLX = range(8)
    C375 = "s[%d:%d]" % (SB, SB + 1)
    ROW0 = "s[%d:%d]" % (SB + 2, SB + 3)           # lanes of row 0 in the four lane groups

    nscr = [0]

    def fmac_bcast(dst, src, mul, lane):
        """dst += (src of row `lane`, broadcast within each 16-lane row) * mul"""
        dpp = "row_newbcast:%d row_mask:0xf bank_mask:0xf" % lane
        if args.plain_fmac:
            tmp = vp(BASE + 32 + 2 * (nscr[0] % 4))
            nscr[0] += 1
            g.valu("v_mov_b64_dpp %s, %s %s" % (tmp, src, dpp), writes=[tmp], dpp_reads=[src])
            g.valu("v_fmac_f64_e32 %s, %s, %s" % (dst, tmp, mul), writes=[dst], reads=[tmp, mul, dst])
        else:
            g.valu("v_fmac_f64_dpp %s, %s, %s %s" % (dst, src, mul, dpp), writes=[dst], reads=[mul, dst], dpp_reads=[src])

    g.salu("s_mov_b32 s%d, 0" % SB)
    g.salu("s_mov_b32 s%d, 0x3fd80000" % (SB + 1))            # 0.375
    g.salu("s_mov_b32 s%d, 0x00010001" % (SB + 2))
    g.salu("s_mov_b32 s%d, 0x00010001" % (SB + 3))
    g.nop(2)                                                   # whatever VALU wrote a[] just before the block

    def head(j):
        """pivot chain of column j as a list of thunks (one instruction each), in dependence order"""
        p = j & 1
        piv, y, t, e, u, rd, nl = (vp(T(p, k)) for k in (PIV, Y, TT, E, U, RD, NL))
        aj = A[j]
        dpp = "row_newbcast:%d row_mask:0xf bank_mask:0xf" % j
        ops = []
        if not args.dpp_rsq:
            ops.append(lambda: g.valu("v_mov_b64_dpp %s, %s %s" % (piv, aj, dpp), writes=[piv], dpp_reads=[aj]))
            ops.append(lambda: g.valu("v_rsq_f64_e32 %s, %s" % (y, piv), writes=[y], reads=[piv], trans=True))
        else:
            ops.append(lambda: g.valu("v_rsq_f64_dpp %s, %s %s" % (y, aj, dpp), writes=[y], dpp_reads=[aj], trans=True))
            ops.append(lambda: g.valu("v_mov_b64_dpp %s, %s %s" % (piv, aj, dpp), writes=[piv], dpp_reads=[aj]))
        ops.append(lambda: g.valu("v_mul_f64 %s, %s, %s" % (t, piv, y), writes=[t], reads=[piv, y]))
        ops.append(lambda: g.valu("v_fma_f64 %s, -%s, %s, 1.0" % (e, t, y), writes=[e], reads=[t, y]))
        ops.append(lambda: g.valu("v_mul_f64 %s, %s, %s" % (u, y, e), writes=[u], reads=[y, e]))
        ops.append(lambda: g.valu("v_fma_f64 %s, %s, %s, 0.5" % (t, e, C375), writes=[t], reads=[e]))
        ops.append(lambda: g.valu("v_fma_f64 %s, %s, %s, %s" % (rd, u, t, y), writes=[rd], reads=[u, t, y]))
        ops.append(lambda: g.valu("v_mul_f64 %s, -%s, %s" % (nl, aj, rd), writes=[nl], reads=[aj, rd]))
        ops.append(lambda: g.valu("v_mul_f64 %s, %s, %s" % (aj, aj, rd), writes=[aj], reads=[aj, rd]))
        # off the chain, ONE schedulable unit (nothing else may run under a narrowed exec): the multiplier is zeroed on rows
        # <= j (they are final in the inverse and "don't care" in the factor), row j of the inverse is scaled by rd
        nk = (j >> 2) + 1
        m16 = (1 << (j + 1)) - 1
        def masked():
            g.salu("s_mov_b32 s%d, 0x%08x" % (SB + 4, m16 | (m16 << 16)))
            g.salu("s_mov_b32 s%d, 0x%08x" % (SB + 5, m16 | (m16 << 16)))
            g.salu("s_mov_b64 exec, s[%d:%d]" % (SB + 4, SB + 5))
            g.valu("v_mov_b64_e32 %s, 0" % nl, writes=[nl])
            g.salu("s_lshl_b64 exec, %s, %d" % (ROW0, j))
            for k in range(nk):
                g.valu("v_mul_f64 %s, %s, %s" % (X[k], X[k], rd), writes=[X[k]], reads=[X[k], rd])
            g.salu("s_mov_b64 exec, -1")
        tail = [masked]
        return ops, tail

    def work(j):
        """everything of column j that is not on the chain to column j + 1: rank-1 updates of columns j + 2 .., the
        inverse's rows; needs nl of column j"""
        p = j & 1
        nl = vp(T(p, NL))
        aj = A[j]
        ops = []
        nk = (j >> 2) + 1
        for c in range(j + 2, 16):
            ops.append(lambda c=c: fmac_bcast(A[c], aj, nl, c))
        for k in range(nk):
            ops.append(lambda k=k: fmac_bcast(X[k], X[k], nl, j))
        return ops

    # column 0's chain has nothing to hide behind
    h, t = head(0)
    for op in h + t:
        op()
    for j in range(16):
        p = j & 1
        nl = vp(T(p, NL))
        w = work(j)
        if j < 15:
            c = j + 1
            fmac_bcast(A[c], A[j], nl, c)
            h, t = head(j + 1)
            # two independent instructions between the write of a[j+1] and its DPP reads, then one chain instruction per
            # two of the column's remaining ones
            seq = []
            w = list(w)
            for _ in range(2):
                if w:
                    seq.append(w.pop(0))
            for op in h:
                seq.append(op)
                for _ in range(2):
                    if w:
                        seq.append(w.pop(0))
            seq += w
            seq += t
            for op in seq:
                op()
        else:
            for op in w:
                op()
    g.nop(4)                                                   # the results feed MFMAs / LDS stores of compiler code

    clob = ["\"v%d\"" % r for r in range(BASE, BASE + (40 if args.plain_fmac else 32))] + ["\"s%d\"" % r for r in range(SB, SB + 6)] + ["\"vcc\"", "\"scc\""]
    out = []
    out.append("// potf2_asm.h -- GENERATED by tools/gen_potf2.py; do not edit.  %d VALU instructions." % g.nvalu)
    out.append("// 16x16 Cholesky + inverse of the factor in the registers of one wave (see the generator for the layout and the")
    out.append("// hazard rules).  a[c]: row i = lane & 15 of column c, mirrored in the four lane groups; on return the lower triangle")
    out.append("// holds L (rows i < c of a[c] are garbage), xs[k] = column 4k + (lane >> 4) of row i of L^-1.")
    out.append("// A pivot that is not positive leaves NaN on the diagonal from that column on (the caller tests L[j][j] > 0).")
    out.append("#pragma once")
    out.append("__device__ __forceinline__ void %s(double (&a)[16], double (&xs)[4], int lane) {" % args.name)
    out.append("    const int vi = lane & 15, vg = lane >> 4;")
    out.append("#pragma unroll")
    out.append("    for (int k = 0; k < 4; ++k) xs[k] = (4 * k + vg == vi) ? 1.0 : 0.0;")
    out.append("    asm volatile(")
    for ln in g.lines:
        out.append("        \"%s\\n\\t\"" % ln)
    ops_out = ", ".join("\"+v\"(a[%d])" % c for c in range(16)) + ", " + ", ".join("\"+v\"(xs[%d])" % k for k in range(4))
    out.append("        : %s" % ops_out)
    out.append("        :")
    out.append("        : %s);" % ", ".join(clob))
    out.append("}")
    sys.stdout.write("\n".join(out) + "\n")
    sys.stderr.write("potf2: %d VALU, %d wait states, %d lines\n" % (g.nvalu, g.idx, len(g.lines)))


if __name__ == "__main__":
    main()
