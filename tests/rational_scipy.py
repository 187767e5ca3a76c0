"""TEST INFRASTRUCTURE (not imported by the product): an independent numpy / scipy implementation of the partial-fraction fit that the
library computes in csrc/rational.hip (lqcd_rational_fit) -- LAPACK SVD and a generalised eigenvalue problem for the poles instead of the
library's Householder / Jacobi SVD and its sign scan.  tests/test_rational.py compares the two.

Partial-fraction approximations of x^(-alpha), 0 < alpha < 1, on a spectral interval -- the coefficients the RHMC path feeds to
the multi-shift solver `shiftedcg` (the reference's general-Nf staggered runs, README.md:112,132,
test/test_Nf2.toml:8, test/test_Nf3.toml:8, obtain theirs from its rational-HMC package).

    x^(-alpha)  ~=  a0 + sum_k  r_k / (x + p_k),      a0 >= 0, r_k > 0,  p_k > 0            for  lam_min <= x <= lam_max

so that  (D'D)^(-alpha) phi ~= a0 phi + sum_k r_k (D'D + p_k)^(-1) phi  is ONE multi-shift solve.  The fit is the AAA algorithm
(Nakatsukasa, Sete, Trefethen 2018: greedy barycentric interpolation + a least-squares weight vector from an SVD), which for
these Stieltjes functions reaches 1e-11 relative accuracy with 10-12 poles -- the pole count of a Remez fit -- in plain double
precision; poles and residues come from the barycentric form (generalised eigenvalue problem, residue formula) and are checked
to be real and of the right sign.  Host-side set-up code (numpy / scipy), nothing here touches the device."""
import numpy as np
import scipy.linalg as sla


def _aaa(F, Z, tol, mmax):
    M = len(Z)
    J = np.arange(M)
    z, f = [], []
    C = np.zeros((M, 0))
    R = np.full(M, F.mean())
    w = None
    err = np.inf
    for _ in range(mmax):
        j = int(np.argmax(np.abs(F - R) / np.abs(F)))
        z.append(Z[j]); f.append(F[j])
        J = J[J != j]
        C = np.column_stack([C, 1.0 / (Z - Z[j] + (Z == Z[j]))])
        A = (F[:, None] * C - C * np.array(f)[None, :])[J]
        w = np.linalg.svd(A, full_matrices=False)[2][-1]
        N, D = C @ (w * np.array(f)), C @ w
        R = F.copy()
        R[J] = N[J] / D[J]
        err = np.abs(R / F - 1.0).max()
        if err < tol:
            break
    return np.array(z), np.array(f), w, err


def inverse_power_partial_fractions(alpha, lam_min, lam_max, tol=1e-10, max_poles=40):
    """Returns (a0, residues r_k, poles p_k, max relative error on [lam_min, lam_max])."""
    if not (0.0 < alpha < 1.0):
        raise ValueError("alpha must lie in (0, 1)")
    if not (0.0 < lam_min < lam_max):
        raise ValueError("need 0 < lam_min < lam_max")
    Z = np.exp(np.linspace(np.log(lam_min), np.log(lam_max), 3000))
    z, f, w, _ = _aaa(Z ** (-alpha), Z, 0.1 * tol, max_poles)
    m = len(z)
    B = np.eye(m + 1); B[0, 0] = 0.0
    E = np.zeros((m + 1, m + 1)); E[0, 1:] = w; E[1:, 0] = 1.0; E[1:, 1:] = np.diag(z)
    pol = sla.eig(E, B, right=False)
    pol = pol[np.isfinite(pol)]
    num = ((w * f)[None, :] / (pol[:, None] - z[None, :])).sum(axis=1)
    dden = -(w[None, :] / (pol[:, None] - z[None, :]) ** 2).sum(axis=1)
    res = num / dden
    a0 = (w * f).sum() / w.sum()
    if np.abs(pol.imag).max() > 1e-12 * np.abs(pol).max() or (pol.real >= 0).any() or (res.real <= 0).any() or a0 < 0:
        raise RuntimeError("rational fit produced poles/residues of the wrong kind; widen the interval or loosen tol")
    poles, res = -pol.real, res.real
    order = np.argsort(poles)
    poles, res = poles[order], res[order]
    # polish a0 and the residues with the poles fixed (the pole/residue form loses a digit or two against the barycentric one):
    # linear least squares on the relative error over the sample set
    Mx = np.column_stack([np.ones_like(Z)] + [1.0 / (Z + pk) for pk in poles]) * (Z ** alpha)[:, None]
    sol = np.linalg.lstsq(Mx, np.ones_like(Z), rcond=None)[0]
    if sol[0] >= 0 and (sol[1:] > 0).all():
        a0, res = sol[0], sol[1:]
    xs = np.exp(np.linspace(np.log(lam_min), np.log(lam_max), 1999))
    err = np.abs(evaluate(a0, res, poles, xs) / xs ** (-alpha) - 1.0).max()
    if err > tol:
        raise RuntimeError(f"rational fit reached {err:.2e}, requested {tol:.2e}")
    return float(a0), res, poles, float(err)


def evaluate(a0, res, poles, x):
    x = np.asarray(x, dtype=np.float64)
    return (a0 + (res[None, :] / (x.reshape(-1, 1) + poles[None, :])).sum(axis=1)).reshape(x.shape)
