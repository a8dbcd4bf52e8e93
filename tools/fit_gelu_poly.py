"""The polynomial GELU of the fused MLP kernel (csrc/common.hpp gelu_erf_poly2): fit and error report.

    gelu(z) = z * Phi(z),   Phi(z) - 1/2 = zc * P(zc^2),   zc = clamp(z, -Z, Z),   P of degree N in s = zc^2

P is a weighted minimax fit (iteratively re-weighted least squares on Chebyshev nodes) of (Phi(z) - 1/2) / z on [0, Z] with the
weight z, i.e. it minimises the absolute error of Phi.  `python tools/fit_gelu_poly.py` re-derives the coefficients and prints the
f32-evaluated error of the committed ones (COEF below = the literals in common.hpp; tests/test_gelu_poly.py checks both)."""
import numpy as np

Z = 4.25
# P(s) = COEF[0] + COEF[1] s + ... + COEF[8] s^8   (the kernel evaluates it by Horner from COEF[8] down, in f32 FMAs)
COEF = (3.989023268e-01, -6.634449214e-02, 9.815969504e-03, -1.108560245e-03, 9.341857367e-05, -5.626413895e-06,
        2.255418963e-07, -5.327728037e-09, 5.564818051e-11)


def phi(z):
    from scipy.special import erf
    return 0.5 * (1.0 + erf(np.asarray(z, dtype=np.float64) / np.sqrt(2.0)))


def gelu_poly_f32(z, coef=COEF, zmax=Z):
    """What the kernel computes, in f32 (numpy's mul + add instead of a fused multiply-add: same to ~1 ulp)."""
    z = np.asarray(z, dtype=np.float32)
    zc = np.clip(z, np.float32(-zmax), np.float32(zmax))
    s = zc * zc
    p = np.full_like(s, np.float32(coef[-1]))
    for c in coef[-2::-1]:
        p = p * s + np.float32(c)
    ph = zc * p + np.float32(0.5)
    return z * ph, ph


def fit(zmax, n, iters=80):
    from numpy.polynomial import chebyshev as C, Polynomial
    u = (np.cos(np.linspace(0, np.pi, 4001)) + 1) / 2
    z = np.sqrt(u) * zmax
    w = np.maximum(z, 1e-3)
    y = np.where(z < 1e-8, 1 / np.sqrt(2 * np.pi), (phi(z) - 0.5) / np.where(z == 0, 1, z))
    A = C.chebvander(2 * u - 1, n)
    wt = np.ones_like(u)
    for _ in range(iters):
        c = np.linalg.lstsq(A * (w * wt)[:, None], y * w * wt, rcond=None)[0]
        r = np.abs((A @ c - y) * w)
        wt *= (r / r.max() + 1e-3) ** 0.5
        wt /= wt.max()
    return Polynomial(C.cheb2poly(c))(Polynomial([-1, 2 / zmax**2])).coef


def report(coef=COEF, zmax=Z):
    z = np.linspace(-12, 12, 2400001)
    g, ph = gelu_poly_f32(z, coef, zmax)
    ref_phi = phi(z)
    ref = z * ref_phi
    rel = np.abs(g - ref) / np.maximum(np.abs(ref), 1e-30)
    return {"max_abs_phi": float(np.abs(ph - ref_phi).max()), "max_abs_gelu": float(np.abs(g - ref).max()),
            "max_rel_gelu_z_gt_0.25": float(rel[z > 0.25].max()), "min_phi": float(ph.min()), "max_phi": float(ph.max())}


if __name__ == "__main__":
    print("committed:", report())
    co = fit(Z, 8)
    print("refit    :", ", ".join(f"{c:.9e}" for c in co))
    print("refit err:", report(tuple(np.float32(co).astype(np.float64)), Z))
