"""Generates the polynomial coefficients of csrc/pmath.h (fp64 evaluation, one rounding to fp32).

    python tools/pmath_coeffs.py

Remez exchange in 200-bit arithmetic (mpmath) for
  log :  atanh(sqrt z) / sqrt z        on z in [0, ((sqrt2 - 1) / (sqrt2 + 1))^2]      (log m = 2 s g(s^2), s = (m - 1) / (m + 1))
  exp :  exp(r)                        on r in [-ln2 / 2, ln2 / 2]
  sin :  sin(sqrt z) / sqrt z          on z in [0, (pi / 4)^2]
  cos :  cos(sqrt z)                   on z in [0, (pi / 4)^2]
  rcp :  1 / t                         on t in [1 + sqrt(1/2), 1 + sqrt 2]              (seed of the Newton reciprocal in pm_log)
and a scan for the magic constant of the inverse-cube-root seed.  Prints C initialisers as hex floats (exact) with the
reached maximum error; the degree of each is the smallest whose error stays below 2^-47 (so that the fp64 result, rounded
once, is the correctly rounded fp32 value except for arguments within ~2^-22 ulp of a rounding boundary)."""
import mpmath as mp

mp.mp.prec = 200


def remez(f, a, b, n, rel=False, iters=30, grid=4000):
    """Coefficients c_0..c_n of the polynomial minimising max |f - p| (or |1 - p / f| with rel) on [a, b]."""
    a, b = mp.mpf(a), mp.mpf(b)
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * i + 1) / (2 * (n + 2))) for i in range(n + 2)][::-1]
    w = (lambda x: 1 / f(x)) if rel else (lambda x: mp.mpf(1))
    best = None
    for _ in range(iters):
        A = mp.matrix(n + 2, n + 2)
        rhs = mp.matrix(n + 2, 1)
        for i, x in enumerate(xs):
            for k in range(n + 1):
                A[i, k] = x ** k
            A[i, n + 1] = (-1) ** i / w(x)
            rhs[i] = f(x)
        sol = mp.lu_solve(A, rhs)
        c = [sol[k] for k in range(n + 1)]
        err = lambda x: w(x) * (f(x) - mp.polyval(c[::-1], x))
        pts = [a + (b - a) * i / grid for i in range(grid + 1)]
        vals = [err(x) for x in pts]
        # local extrema of the error, alternating in sign
        ext = []
        for i in range(grid + 1):
            l = vals[i - 1] if i > 0 else None
            r = vals[i + 1] if i < grid else None
            v = vals[i]
            if (l is None or abs(v) >= abs(l)) and (r is None or abs(v) >= abs(r)):
                if ext and mp.sign(vals[ext[-1]]) == mp.sign(v):
                    if abs(v) > abs(vals[ext[-1]]):
                        ext[-1] = i
                else:
                    ext.append(i)
        emax = max(abs(v) for v in vals)
        best = (c, emax)
        if len(ext) < n + 2:
            break
        while len(ext) > n + 2:                      # drop the smaller end
            if abs(vals[ext[0]]) < abs(vals[ext[-1]]):
                ext.pop(0)
            else:
                ext.pop()
        new = [pts[i] for i in ext]
        if all(abs(p - q) < (b - a) * mp.mpf(10) ** -12 for p, q in zip(new, xs)):
            break
        xs = new
    return best


def smallest(f, a, b, target, rel=False, lo=2, hi=16):
    for n in range(lo, hi):
        c, e = remez(f, a, b, n, rel)
        if e < target:
            return n, c, e
    raise RuntimeError("no degree reaches the target")


def show(name, c, e):
    print("// %s: degree %d, max error 2^%.1f" % (name, len(c) - 1, float(mp.log(e, 2))))
    print("    " + ", ".join(float(x).hex() for x in c))


if __name__ == "__main__":
    T = mp.mpf(2) ** -47
    s = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
    g = lambda z: mp.atanh(mp.sqrt(z)) / mp.sqrt(z) if z > 0 else mp.mpf(1)
    show("log: atanh(sqrt z)/sqrt z", *smallest(g, 0, s * s, T)[1:])
    show("exp(r)", *smallest(mp.exp, -mp.log(2) / 2, mp.log(2) / 2, T, rel=True)[1:])
    q = (mp.pi / 4) ** 2
    show("sin(sqrt z)/sqrt z", *smallest(lambda z: mp.sin(mp.sqrt(z)) / mp.sqrt(z) if z > 0 else mp.mpf(1), 0, q, T)[1:])
    show("cos(sqrt z)", *smallest(lambda z: mp.cos(mp.sqrt(z)), 0, q, T, rel=True)[1:])
    show("1/t seed", *remez(lambda t: 1 / t, 1 + mp.sqrt(mp.mpf(1) / 2), 1 + mp.sqrt(2), 2, rel=True))
