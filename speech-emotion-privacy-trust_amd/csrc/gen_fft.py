#!/usr/bin/env python3
"""Generate fully unrolled in-register complex FFTs (forward, e^{-2*pi*i*nk/R}) for the
mel kernel's two-pass mixed-radix STFT.  Output: fft_reg_gen.h (committed; re-generate with
`python gen_fft.py > fft_reg_gen.h`).

Each emitted function
    template<> struct FftReg<R> { static __device__ void run(float (&re)[R], float (&im)[R]); }
transforms in place, natural order in, natural order out.  All indices are literals so
hipcc keeps the arrays in VGPRs; all twiddles are float literals (rounded from float64),
so they become inline/SGPR constants, not table loads.

Decomposition: decimation in time, R = r * m with r in {4, 5, 2} (radix-4 first), recursing
on the m-point sub-transforms.  Sizes needed: n_fft 800 -> 20x20, 1600 -> 40x20,
1024 -> 16x32, 400 (MFCC) -> 10x20.
"""
import math
import sys

SIZES = [8, 10, 16, 20, 25, 32, 40, 50]


class Emitter:
    def __init__(self):
        self.lines = []
        self.n = 0

    def tmp(self):
        self.n += 1
        return f"t{self.n}"

    def emit(self, expr):
        v = self.tmp()
        self.lines.append(f"    const float {v} = {expr};")
        return v


def lit(x):
    if x == 0.0:
        return "0.0f"
    return f"{x:.9e}f"


def cmul_const(E, a, wr, wi):
    """(ar + i ai) * (wr + i wi) with literal twiddle; special-cases the trivial ones."""
    ar, ai = a
    eps = 1e-15
    if abs(wr - 1) < eps and abs(wi) < eps:
        return a
    if abs(wr + 1) < eps and abs(wi) < eps:
        return (E.emit(f"-{ar}"), E.emit(f"-{ai}"))
    if abs(wr) < eps and abs(wi + 1) < eps:  # * -i
        return (ai, E.emit(f"-{ar}"))
    if abs(wr) < eps and abs(wi - 1) < eps:  # * +i
        return (E.emit(f"-{ai}"), ar)
    r = E.emit(f"{ar} * {lit(wr)} - {ai} * {lit(wi)}")
    i = E.emit(f"{ar} * {lit(wi)} + {ai} * {lit(wr)}")
    return (r, i)


def bfly2(E, x):
    (ar, ai), (br, bi) = x
    return [(E.emit(f"{ar} + {br}"), E.emit(f"{ai} + {bi}")),
            (E.emit(f"{ar} - {br}"), E.emit(f"{ai} - {bi}"))]


def bfly4(E, x):
    (ar, ai), (br, bi), (cr, ci), (dr, di) = x
    t0r, t0i = E.emit(f"{ar} + {cr}"), E.emit(f"{ai} + {ci}")
    t1r, t1i = E.emit(f"{ar} - {cr}"), E.emit(f"{ai} - {ci}")
    t2r, t2i = E.emit(f"{br} + {dr}"), E.emit(f"{bi} + {di}")
    # t3 = (b - d) * (-i)  -> (im, -re)
    t3r, t3i = E.emit(f"{bi} - {di}"), E.emit(f"{dr} - {br}")
    return [(E.emit(f"{t0r} + {t2r}"), E.emit(f"{t0i} + {t2i}")),
            (E.emit(f"{t1r} + {t3r}"), E.emit(f"{t1i} + {t3i}")),
            (E.emit(f"{t0r} - {t2r}"), E.emit(f"{t0i} - {t2i}")),
            (E.emit(f"{t1r} - {t3r}"), E.emit(f"{t1i} - {t3i}"))]


def bfly5(E, x):
    (ar, ai), (br, bi), (cr, ci), (dr, di), (er, ei) = x
    c1, c2 = math.cos(2 * math.pi / 5), math.cos(4 * math.pi / 5)
    s1, s2 = math.sin(2 * math.pi / 5), math.sin(4 * math.pi / 5)
    t1r, t1i = E.emit(f"{br} + {er}"), E.emit(f"{bi} + {ei}")
    t2r, t2i = E.emit(f"{cr} + {dr}"), E.emit(f"{ci} + {di}")
    t3r, t3i = E.emit(f"{br} - {er}"), E.emit(f"{bi} - {ei}")
    t4r, t4i = E.emit(f"{cr} - {dr}"), E.emit(f"{ci} - {di}")
    x0 = (E.emit(f"{ar} + {t1r} + {t2r}"), E.emit(f"{ai} + {t1i} + {t2i}"))
    m1r = E.emit(f"{ar} + {lit(c1)} * {t1r} + {lit(c2)} * {t2r}")
    m1i = E.emit(f"{ai} + {lit(c1)} * {t1i} + {lit(c2)} * {t2i}")
    m2r = E.emit(f"{ar} + {lit(c2)} * {t1r} + {lit(c1)} * {t2r}")
    m2i = E.emit(f"{ai} + {lit(c2)} * {t1i} + {lit(c1)} * {t2i}")
    n1r = E.emit(f"{lit(s1)} * {t3r} + {lit(s2)} * {t4r}")
    n1i = E.emit(f"{lit(s1)} * {t3i} + {lit(s2)} * {t4i}")
    n2r = E.emit(f"{lit(s2)} * {t3r} - {lit(s1)} * {t4r}")
    n2i = E.emit(f"{lit(s2)} * {t3i} - {lit(s1)} * {t4i}")
    # X1 = m1 - i n1, X4 = m1 + i n1, X2 = m2 - i n2, X3 = m2 + i n2;  -i*(nr + i ni) = ni - i nr
    x1 = (E.emit(f"{m1r} + {n1i}"), E.emit(f"{m1i} - {n1r}"))
    x4 = (E.emit(f"{m1r} - {n1i}"), E.emit(f"{m1i} + {n1r}"))
    x2 = (E.emit(f"{m2r} + {n2i}"), E.emit(f"{m2i} - {n2r}"))
    x3 = (E.emit(f"{m2r} - {n2i}"), E.emit(f"{m2i} + {n2r}"))
    return [x0, x1, x2, x3, x4]


BFLY = {2: bfly2, 4: bfly4, 5: bfly5}


def pick_radix(n):
    for r in (4, 5, 2):
        if n % r == 0:
            return r
    raise ValueError(f"unsupported size {n}")


def fft(E, x):
    """x: list of (re, im) symbol pairs, natural order -> list of outputs, natural order."""
    n = len(x)
    if n == 1:
        return x
    r = pick_radix(n)
    m = n // r
    subs = [fft(E, x[j::r]) for j in range(r)]
    out = [None] * n
    for k2 in range(m):
        ins = []
        for j in range(r):
            ang = -2.0 * math.pi * (j * k2) / n
            ins.append(cmul_const(E, subs[j][k2], math.cos(ang), math.sin(ang)))
        ys = BFLY[r](E, ins)
        for q in range(r):
            out[k2 + m * q] = ys[q]
    return out


def gen(R):
    E = Emitter()
    x = [(f"re[{i}]", f"im[{i}]") for i in range(R)]
    # copy inputs to scalars first so in-place stores cannot alias later reads
    xs = [(E.emit(a), E.emit(b)) for a, b in x]
    out = fft(E, xs)
    body = "\n".join(E.lines)
    stores = "\n".join(f"    re[{i}] = {o[0]}; im[{i}] = {o[1]};" for i, o in enumerate(out))
    return (f"template <> struct FftReg<{R}> {{\n"
            f"  static __device__ __forceinline__ void run(float (&re)[{R}], float (&im)[{R}]) {{\n"
            f"{body}\n{stores}\n  }}\n}};\n")


def main():
    print("// GENERATED by gen_fft.py -- do not edit.  In-register forward complex FFTs.")
    print("#pragma once")
    print("template <int R> struct FftReg;\n")
    for R in SIZES:
        print(gen(R))


if __name__ == "__main__":
    main()
