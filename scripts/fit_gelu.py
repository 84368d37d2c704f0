import numpy as np
from scipy.special import erfc
from scipy.optimize import least_squares
# target: h(t) = t * erfc(t)/2 on [0, T]; model: t * 2^(t*R(t) - 1); R poly degree d. Minimise max abs error of h (=> error of GELU/sqrt2).
def fit(d, T=4.0, n=4001):
    t = np.linspace(1e-6, T, n)
    target = np.log2(erfc(t)) / t          # R(t) exact
    w = t * erfc(t) / 2 * t * np.log(2)    # d h / d R  = h * t * ln2 -> weight for linearised error
    # weighted LS init
    V = np.vander(t, d + 1)
    c = np.linalg.lstsq(V * w[:, None], target * w, rcond=None)[0]
    def resid(c):
        R = np.polyval(c, t)
        return t * np.exp2(t * R - 1) - t * erfc(t) / 2
    # minimax-ish via p-norm
    for p in (2, 4, 8, 16, 32):
        r = least_squares(lambda c: np.sign(resid(c)) * np.abs(resid(c)) ** (p / 2), c, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        c = r.x
    e = resid(c)
    # f32 evaluation
    tf = t.astype(np.float32); cf = c.astype(np.float32)
    R = np.full_like(tf, cf[0])
    for k in cf[1:]: R = (R * tf + k).astype(np.float32)
    hf = tf * np.exp2((R * tf - np.float32(1)).astype(np.float32)).astype(np.float32)
    ef = hf.astype(np.float64) - t * erfc(t) / 2
    return c, np.abs(e).max(), np.abs(ef).max()
for d in (4, 5, 6, 7):
    c, e, ef = fit(d)
    print(d, "max err exact-arith %.3g  f32-eval %.3g" % (e, ef), " (GELU abs err = sqrt2 x this)")
    print("   coeffs", ", ".join("%.9e" % x for x in c))
