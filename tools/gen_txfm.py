#!/usr/bin/env python3
"""gen_txfm.py -- generates the straight-line 1-D AV1 transform kernels (DCT 4..64, ADST 8/16; forward and inverse; the legacy 32-point inverse ADST).

The AV1 DCT/ADST are fixed-point butterfly networks with a rounding shift after every rotation
(half_btf, Source/Lib/Codec/inv_transforms.h:260-285), so a bit-exact implementation must realise the same flow graph.
Instead of transcribing the reference's 7000 unrolled lines (transforms.c:50-2236, inv_transforms.c:94-2361) this tool
states the graph's *recursive structure* and derives everything else:

  fdct(N)   = mirror butterfly -> fdct(N/2) on the sums | odd(N/2) on the differences, bit-reversed output
  odd(M)    = log2(M)-1 levels of { rotation of the middle half of every block against its mirror image,
              butterflies inside alternating +/- blocks of half the size } followed by the final rotation stage
              with angles 64 - u(2*brev(i)+1)
  fadst(N)  = signed input permutation -> levels of { pi/4, pi/8.. rotations of the upper half of each group,
              stride-2^t butterflies } -> final rotations (angles (64/N)/2 + (128/N) i) -> output permutation
  inverse   = the transposed graph run backwards (every rotation matrix transposed, every add/sub followed by the
              stage clamp of inv_transforms.c:86-92) -- checked against svt_av1_idct*/iadst* by tests.

  iadst32   = NOT a transposed graph: the reference's av1_iadst32_new (inv_transforms.c:1119-1552; a transform type AV1 never signals for a 32-point
              dimension, but svt_av1_inv_txfm2d_add_{16x32,32x16,8x32,32x8,32x64,64x32}_c accept it and the reference's own InvTxfm2dAddTest fixture feeds it,
              test/InvTxfm2dAsmTest.cc:755-775) is the FORWARD network shape fadst(32) with clamp_buf over the whole array after every stage -- the signed
              input permutation and the rotation outputs are clamped too, not only the adds.

Outputs (committed, regenerate with `python tools/gen_txfm.py`):
  oracle/oracle_txfm1d_gen.h            plain C, used by the test oracle
  svt-av1-psy_amd/csrc/txfm1d_gen.h     HIP device code: in-place on a register array, cos_bit as template constant
Both are validated bit-exact against the real reference's 1-D functions in tests/test_oracle_pin_txfm.py.
"""
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def brev(v, bits):
    r = 0
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def ilog2(n):
    return n.bit_length() - 1


# ------------------------------------------------------------------------------------------------------------------
# IR: a program is a list of passes; a pass is a list of pair-ops acting on slots of the working array:
#   ("rot", j, p, wjj, wjp, wpj, wpp): new[j] = R(wjj*old[j] + wjp*old[p]); new[p] = R(wpj*old[j] + wpp*old[p])
#       weights are signed cos indices: +k = cospi[k], -k = -cospi[k]  (k in 1..63)
#   ("add", x, y, sxx, sxy, syx, syy): new[x] = sxx*old[x] + sxy*old[y]; new[y] = syx*old[x] + syy*old[y], s = +-1
#   ("perm", [(src, sign), ...]): new[i] = sign * old[src]
# ------------------------------------------------------------------------------------------------------------------
def fdct_prog(N):
    passes = []

    def bfly(idx, typ_a=True):
        ops = []
        S = len(idx)
        for i in range(S // 2):
            x, y = idx[i], idx[S - 1 - i]
            ops.append(("add", x, y, 1, 1, 1, -1) if typ_a else ("add", x, y, -1, 1, 1, 1))
        return ops

    def odd(idx):
        M = len(idx)
        m = ilog2(M)
        for t in range(1, m):
            ops = []
            if t == 1:
                for j in range(M // 4, M // 2):
                    p = M - 1 - j
                    ops.append(("rot", idx[j], idx[p], -32, 32, 32, 32))
            else:
                nblk, S, u = 1 << (t - 2), M >> (t - 1), 64 >> t
                for b in range(nblk):
                    a = u * (2 * brev(b, t - 1) + 1)
                    lo = b * S
                    for j in range(lo + S // 4, lo + S // 2):      # first quarter of the middle half
                        p = M - 1 - j
                        ops.append(("rot", idx[j], idx[p], -a, 64 - a, 64 - a, a))
                    for j in range(lo + S // 2, lo + 3 * S // 4):  # second quarter
                        p = M - 1 - j
                        ops.append(("rot", idx[j], idx[p], -(64 - a), -a, -a, 64 - a))
            passes.append(ops)
            ops = []
            S2 = M >> t
            for blk in range(1 << t):
                ops += bfly(idx[blk * S2:(blk + 1) * S2], typ_a=(blk % 2 == 0))
            passes.append(ops)
        u = 64 // (2 * M)
        ops = []
        for i in range(M // 2):
            p = M - 1 - i
            a = u * (2 * brev(i, m) + 1)
            ops.append(("rot", idx[i], idx[p], 64 - a, a, -a, 64 - a))
        passes.append(ops)

    def rec(idx):
        n = len(idx)
        if n == 2:
            passes.append([("rot", idx[0], idx[1], 32, 32, 32, -32)])
            return
        passes.append(bfly(idx, True))
        rec(idx[:n // 2])
        odd(idx[n // 2:])

    rec(list(range(N)))
    bits = ilog2(N)
    passes.append([("perm", [(brev(i, bits), 1) for i in range(N)])])
    return passes


ADST_IN = {  # signed input permutation of the forward ADST (slot i <- sign * x[src]); rest of the network is generic
    8: [(0, 1), (7, -1), (3, -1), (4, 1), (1, -1), (6, 1), (2, 1), (5, -1)],
    16: [(0, 1), (15, -1), (7, -1), (8, 1), (3, -1), (12, 1), (4, 1), (11, -1), (1, -1), (14, 1), (6, 1), (9, -1), (2, 1), (13, -1),
         (5, -1), (10, 1)],
}
ADST_OUT = {8: [1, 6, 3, 4, 5, 2, 7, 0], 16: [1, 14, 3, 12, 5, 10, 7, 8, 9, 6, 11, 4, 13, 2, 15, 0]}
# the same two tables continued to N = 32 (slot i <- sign * x[src] is "even bit-reversed positions from the front, odd ones mirrored from the back, signs in the
# +--+ -++- pattern of the smaller sizes"; the output takes odd slots ascending interleaved with even slots descending)
ADST_IN[32] = [(0, 1), (31, -1), (15, -1), (16, 1), (7, -1), (24, 1), (8, 1), (23, -1), (3, -1), (28, 1), (12, 1), (19, -1), (4, 1), (27, -1), (11, -1), (20, 1),
               (1, -1), (30, 1), (14, 1), (17, -1), (6, 1), (25, -1), (9, -1), (22, 1), (2, 1), (29, -1), (13, -1), (18, 1), (5, -1), (26, 1), (10, 1), (21, -1)]
ADST_OUT[32] = [x for i in range(16) for x in (2 * i + 1, 30 - 2 * i)]


def fadst_prog(N):
    passes = [[("perm", ADST_IN[N])]]
    n = ilog2(N)
    for t in range(1, n):
        half, grp = 1 << t, 2 << t
        u = 64 >> t
        ops = []
        for g0 in range(0, N, grp):
            pairs = [(g0 + half + 2 * i, g0 + half + 2 * i + 1) for i in range(half // 2)]
            if t == 1:
                ops.append(("rot", pairs[0][0], pairs[0][1], 32, 32, 32, -32))
            else:
                q = len(pairs) // 2
                for i, (x, y) in enumerate(pairs):
                    a = u * (1 + 4 * (i % q))
                    if i < q:
                        ops.append(("rot", x, y, a, 64 - a, 64 - a, -a))
                    else:
                        ops.append(("rot", x, y, -(64 - a), a, a, 64 - a))
        passes.append(ops)
        ops = []
        for g0 in range(0, N, grp):
            for i in range(half):
                ops.append(("add", g0 + i, g0 + i + half, 1, 1, 1, -1))
        passes.append(ops)
    # the reference orders these as rot(t=1), add(stride 2), rot(t=2), add(stride 4) ...; our loop emitted
    # rot(t) then add(stride 2^t) which is the same sequence.
    ops = []
    for i in range(N // 2):
        a = (64 // N) // 2 + (128 // N) * i
        ops.append(("rot", 2 * i, 2 * i + 1, a, 64 - a, 64 - a, -a))
    passes.append(ops)
    passes.append([("perm", [(s, 1) for s in ADST_OUT[N]])])
    return passes


def clamp_all_prog(passes):
    """the forward-shaped network with the stage clamp applied to every value a stage produces (legacy inverse ADST32)"""
    out = []
    for ops in passes:
        if ops and ops[0][0] == "perm":
            out.append([("permc", ops[0][1])])
            continue
        out.append([(("rotc",) + op[1:]) if op[0] == "rot" else (("addc",) + op[1:]) for op in ops])
    return out


def transpose_prog(passes, N):
    out = []
    for ops in reversed(passes):
        if ops and ops[0][0] == "perm":
            perm = ops[0][1]
            inv = [None] * N
            for i, (src, sign) in enumerate(perm):
                inv[src] = (i, sign)
            out.append([("perm", inv)])
            continue
        new = []
        for op in ops:
            if op[0] == "rot":
                _, j, p, wjj, wjp, wpj, wpp = op
                new.append(("rot", j, p, wjj, wpj, wjp, wpp))
            else:
                _, x, y, sxx, sxy, syx, syy = op
                new.append(("addc", x, y, sxx, syx, sxy, syy))  # inverse adds are clamped
        out.append(new)
    return out


# ------------------------------------------------------------------------------------------------------------------
def emit(name, passes, N, lang, inverse):
    """lang 'c': in/out pointer API; lang 'hip': in-place on int32_t (&v)[N]."""
    lines = []
    cur = ["%s[%d]" % ("in" if lang == "c" else "v", i) for i in range(N)]
    tmp = [0]

    def nt():
        tmp[0] += 1
        return "t%d" % tmp[0]

    def w(k):
        return ("C(%d)" % k) if k > 0 else ("-C(%d)" % -k)

    def sg(s, v):
        return v if s > 0 else "-" + v

    for ops in passes:
        new = list(cur)
        for op in ops:
            if op[0] == "perm":
                new = [cur[src] if sign > 0 else "NEG(%s)" % cur[src] for (src, sign) in op[1]]
            elif op[0] == "permc":  # a negated value is clamped again: -(-2^(b-1)) is out of range
                new = [cur[src] if sign > 0 else "CL(NEG(%s))" % cur[src] for (src, sign) in op[1]]
            elif op[0] in ("rot", "rotc"):
                _, j, p, wjj, wjp, wpj, wpp = op
                a, b = nt(), nt()
                fmt = "const int32_t %s = CL(HB(%s, %s, %s, %s));" if op[0] == "rotc" else "const int32_t %s = HB(%s, %s, %s, %s);"
                lines.append(fmt % (a, w(wjj), cur[j], w(wjp), cur[p]))
                lines.append(fmt % (b, w(wpj), cur[j], w(wpp), cur[p]))
                new[j], new[p] = a, b
            else:
                kind, x, y, sxx, sxy, syx, syy = op
                a, b = nt(), nt()
                def lin(sa, sb):
                    if sa > 0 and sb > 0:
                        return "ADD(%s, %s)" % (cur[x], cur[y])
                    if sa > 0:
                        return "SUB(%s, %s)" % (cur[x], cur[y])
                    assert sb > 0
                    return "SUB(%s, %s)" % (cur[y], cur[x])
                ea, eb = lin(sxx, sxy), lin(syx, syy)
                if kind == "addc":
                    ea, eb = "CL(%s)" % ea, "CL(%s)" % eb
                lines.append("const int32_t %s = %s;" % (a, ea))
                lines.append("const int32_t %s = %s;" % (b, eb))
                new[x], new[y] = a, b
        cur = new
    if lang == "c":
        sig = "static void %s(const int32_t *in, int32_t *out, int cos_bit%s) {" % (name, ", int clamp_bit" if inverse else "")
        body = ["    const int32_t *cospi = o_cospi(cos_bit);", "    (void)cospi;"]
        if inverse:
            body.append("    const int64_t cl_hi = ((int64_t)1 << (clamp_bit - 1)) - 1, cl_lo = -((int64_t)1 << (clamp_bit - 1));")
        body += ["    " + ln for ln in lines]
        body += ["    out[%d] = %s;" % (i, cur[i]) for i in range(N)]
    else:
        sig = "template <int CB> __device__ __forceinline__ void %s(int32_t (&v)[%d]%s) {" % (name, N, ", const int32_t cl_lo, const int32_t cl_hi" if inverse else "")
        body = ["    " + ln for ln in lines]
        outs = ["o%d" % i for i in range(N)]
        body += ["    const int32_t %s = %s;" % (outs[i], cur[i]) for i in range(N)]
        body += ["    v[%d] = %s;" % (i, outs[i]) for i in range(N)]
    return "\n".join([sig] + body + ["}", ""])


def main():
    progs = []
    for N in (4, 8, 16, 32, 64):
        f = fdct_prog(N)
        progs.append(("fdct%d" % N, f, N, False))
        progs.append(("idct%d" % N, transpose_prog(f, N), N, True))
    for N in (8, 16):
        f = fadst_prog(N)
        progs.append(("fadst%d" % N, f, N, False))
        progs.append(("iadst%d" % N, transpose_prog(f, N), N, True))
    progs.append(("iadst32", clamp_all_prog(fadst_prog(32)), 32, True))

    cos_tabs = [[int(round(math.cos(math.pi * j / 128.0) * (1 << bit))) for j in range(64)] for bit in range(10, 17)]
    head = "// GENERATED by tools/gen_txfm.py -- do not edit.  1-D AV1 DCT/ADST flow graphs (see the generator for the derivation).\n"
    c = [head, "#ifndef ORACLE_TXFM1D_GEN_H", "#define ORACLE_TXFM1D_GEN_H", "#include <stdint.h>",
         "static const int32_t o_cospi_tab[7][64] = {"]
    c += ["    {" + ", ".join(str(v) for v in row) + "}," for row in cos_tabs]
    c += ["};", "static inline const int32_t *o_cospi(int bit) { return o_cospi_tab[bit - 10]; }",
          "/* half_btf, Source/Lib/Codec/inv_transforms.h:260-285: 32-bit wrapping products, 64-bit sum, round, shift */",
          "static inline int32_t o_half_btf(int32_t w0, int32_t in0, int32_t w1, int32_t in1, int bit) {",
          "    int64_t r = (int64_t)(int32_t)((uint32_t)w0 * (uint32_t)in0) + (int64_t)(int32_t)((uint32_t)w1 * (uint32_t)in1);",
          "    return (int32_t)((r + ((int64_t)1 << (bit - 1))) >> bit);", "}",
          "#define C(k) cospi[k]", "#define HB(w0, a, w1, b) o_half_btf(w0, a, w1, b, cos_bit)",
          "#define ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))", "#define SUB(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))",
          "#define NEG(a) ((int32_t)(0u - (uint32_t)(a)))",
          "#define CL(x) ((int32_t)((int64_t)(x) < cl_lo ? cl_lo : ((int64_t)(x) > cl_hi ? cl_hi : (int64_t)(x))))", ""]
    for (name, p, N, inv) in progs:
        c.append(emit("o_" + name, p, N, "c", inv))
    c += ["#undef C", "#undef HB", "#undef CL", "#undef ADD", "#undef SUB", "#undef NEG", "#endif"]
    open(os.path.join(ROOT, "oracle", "oracle_txfm1d_gen.h"), "w").write("\n".join(c) + "\n")

    h = [head, "#pragma once", "#include <stdint.h>", "namespace txfm1d {",
         "// cospi[j] = round(cos(pi*j/128) * 2^bit), bit = 10..13 (svt_aom_eb_av1_cospi_arr_data, inv_transforms.c:3196)",
         "__device__ constexpr int32_t kCospi[4][64] = {"]
    h += ["    {" + ", ".join(str(v) for v in row) + "}," for row in cos_tabs[:4]]
    h += ["};",
          "// half_btf (inv_transforms.h:260-285).  The C code forms each product in int32 and sums in int64; the AV1 stage ranges",
          "// (cos_bit is lowered exactly so that |w * in| < 2^31 at every stage, transforms.h:47-50 / range_check_buf) guarantee the int32",
          "// products never overflow for any residual within the bit depth, i.e. wherever the C expression is defined.  There the exact",
          "// 64-bit form below is identical and costs two v_mad_i64_i32 + one 64-bit shift instead of 2 mul + sign-extensions + carries.",
          "template <int CB> __device__ __forceinline__ int32_t half_btf(const int32_t w0, const int32_t a, const int32_t w1, const int32_t b) {",
          "    const int64_t r = (int64_t)w0 * (int64_t)a + (int64_t)w1 * (int64_t)b + ((int64_t)1 << (CB - 1));",
          "    return (int32_t)(r >> CB);", "}",
          "__device__ __forceinline__ int32_t clamp_i32(const int32_t x, const int32_t lo, const int32_t hi) { const int32_t t = x > lo ? x : lo; return t < hi ? t : hi; } // v_max_i32 + v_min_i32 (lo <= hi)",
          "#define C(k) kCospi[CB - 10][k]", "#define HB(w0, a, w1, b) half_btf<CB>(w0, a, w1, b)",
          "// inverse adds: the C code adds in int32 (wrapping) and then clamps the int32 value (inv_transforms.c:86-92)",
          "#define ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))", "#define SUB(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))",
          "#define NEG(a) ((int32_t)(0u - (uint32_t)(a)))", "#define CL(x) clamp_i32(x, cl_lo, cl_hi)", ""]
    for (name, p, N, inv) in progs:
        h.append(emit(name, p, N, "hip", inv))
    h += ["#undef C", "#undef HB", "#undef CL", "#undef ADD", "#undef SUB", "#undef NEG", "} // namespace txfm1d"]
    open(os.path.join(ROOT, "svt-av1-psy_amd", "csrc", "txfm1d_gen.h"), "w").write("\n".join(h) + "\n")
    print("generated", len(progs), "1-D transforms")


if __name__ == "__main__":
    main()
