// rational.hip -- the host-side numerics of the general-Nf (RHMC) pseudofermion action, inside the library so that every binding gets them
// through the C ABI: partial fractions of x^(-alpha) on a spectral interval (lqcd_rational_fit), the Lanczos estimate of the spectrum of
// D^+D (lqcd_estimate_spectrum) and the FermiAction handle (lqcd_action_*) that owns interval, fits and refits.
// Reference interface: FermiAction(D, Dict("Nf" => n)) /root/reference/src/system/universe.jl:106-110,138; the staggered Nf = 2 / 3 runs of
// test/test_Nf2.toml:8, test/test_Nf3.toml:8, test/runtests.jl:114-130; README.md:112,132 (RHMC).  The reference's package brings Remez tables
// for those runs; here the coefficients are FITTED at construction (AAA: Nakatsukasa, Sete, Trefethen 2018 -- greedy barycentric interpolation,
// weights = smallest right singular vector of the Loewner matrix), which reaches 1e-12 relative accuracy with the pole count of a Remez fit in
// plain double precision.  No LAPACK: Householder QR + one-sided Jacobi SVD on the (at most 40 x 40) triangular factor.
#include "ops_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace lqcd;

namespace {

// ---------------------------------------------------------------------------------- small dense linear algebra (column-major, rows >= n)
// A (rows x n) -> R in its upper triangle (Householder reflections applied in place); b (rows, may be null) <- Q^T b
void householder_qr(std::vector<double>& A, int rows, int n, double* b) {
    std::vector<double> v(rows);
    for (int k = 0; k < n; k++) {
        double* ak = &A[(size_t)k * rows];
        double s = 0.0;
        for (int i = k; i < rows; i++) s += ak[i] * ak[i];
        const double nrm = std::sqrt(s);
        if (nrm == 0.0) continue;
        const double alpha = ak[k] > 0 ? -nrm : nrm;
        for (int i = k; i < rows; i++) v[i] = ak[i];
        v[k] -= alpha;
        const double vv = s - ak[k] * ak[k] + v[k] * v[k];
        if (vv == 0.0) continue;
        const double f = 2.0 / vv;
        for (int j = k + 1; j < n; j++) {
            double* aj = &A[(size_t)j * rows];
            double d = 0.0;
            for (int i = k; i < rows; i++) d += v[i] * aj[i];
            d *= f;
            for (int i = k; i < rows; i++) aj[i] -= d * v[i];
        }
        if (b) {
            double d = 0.0;
            for (int i = k; i < rows; i++) d += v[i] * b[i];
            d *= f;
            for (int i = k; i < rows; i++) b[i] -= d * v[i];
        }
        ak[k] = alpha;
        for (int i = k + 1; i < rows; i++) ak[i] = 0.0;
    }
}

// one-sided (Hestenes) Jacobi SVD of the n x n matrix G (column-major): on return the columns of G are U_j sigma_j, V holds the right
// singular vectors, sig the singular values (unsorted)
void jacobi_svd(std::vector<double>& G, int n, std::vector<double>& V, std::vector<double>& sig) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 80; sweep++) {
        bool rotated = false;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double* gp = &G[(size_t)p * n];
                double* gq = &G[(size_t)q * n];
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < n; i++) { a += gp[i] * gp[i]; b += gq[i] * gq[i]; c += gp[i] * gq[i]; }
                if (c == 0.0 || std::fabs(c) <= 1e-16 * std::sqrt(a * b)) continue;
                rotated = true;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; i++) {
                    const double x = gp[i], y = gq[i];
                    gp[i] = cs * x - sn * y;
                    gq[i] = sn * x + cs * y;
                }
                double* vp = &V[(size_t)p * n];
                double* vq = &V[(size_t)q * n];
                for (int i = 0; i < n; i++) {
                    const double x = vp[i], y = vq[i];
                    vp[i] = cs * x - sn * y;
                    vq[i] = sn * x + cs * y;
                }
            }
        if (!rotated) break;
    }
    sig.assign(n, 0.0);
    for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += G[(size_t)j * n + i] * G[(size_t)j * n + i];
        sig[j] = std::sqrt(s);
    }
}

// right singular vector of the smallest singular value of A (rows x n); A is destroyed
void min_right_singular_vector(std::vector<double>& A, int rows, int n, std::vector<double>& w) {
    householder_qr(A, rows, n, nullptr);
    std::vector<double> R((size_t)n * n, 0.0), V, sig;
    for (int j = 0; j < n; j++)
        for (int i = 0; i <= j; i++) R[(size_t)j * n + i] = A[(size_t)j * rows + i];
    jacobi_svd(R, n, V, sig);
    int jm = 0;
    for (int j = 1; j < n; j++)
        if (sig[j] < sig[jm]) jm = j;
    w.assign(V.begin() + (size_t)jm * n, V.begin() + (size_t)(jm + 1) * n);
}

// minimum-norm least-squares solution of A x = b (rows x n) through QR + SVD of R, singular values below rcond * sigma_max dropped
void lstsq(std::vector<double>& A, int rows, int n, std::vector<double>& b, std::vector<double>& x) {
    householder_qr(A, rows, n, b.data());
    std::vector<double> R((size_t)n * n, 0.0), V, sig;
    for (int j = 0; j < n; j++)
        for (int i = 0; i <= j; i++) R[(size_t)j * n + i] = A[(size_t)j * rows + i];
    jacobi_svd(R, n, V, sig);
    const double smax = *std::max_element(sig.begin(), sig.end());
    const double cut = 2.220446049250313e-16 * std::max(rows, n) * smax;
    x.assign(n, 0.0);
    for (int j = 0; j < n; j++) {
        if (sig[j] <= cut) continue;
        double d = 0;          // (U_j . c) / sigma_j with U_j = R_j / sigma_j
        for (int i = 0; i < n; i++) d += R[(size_t)j * n + i] * b[i];
        d /= sig[j] * sig[j];
        for (int i = 0; i < n; i++) x[i] += d * V[(size_t)j * n + i];
    }
}

// extreme eigenvalues of the symmetric tridiagonal (a_0..a_{n-1}; b_0..b_{n-2}) by Sturm-sequence bisection
int sturm_count(const std::vector<double>& a, const std::vector<double>& b, double x) {     // eigenvalues < x
    int cnt = 0;
    double d = 1.0;
    for (size_t i = 0; i < a.size(); i++) {
        const double off = i ? b[i - 1] * b[i - 1] : 0.0;
        d = a[i] - x - (i ? off / d : 0.0);
        if (d == 0.0) d = 1e-300;
        if (d < 0) cnt++;
    }
    return cnt;
}
double tridiag_eigenvalue(const std::vector<double>& a, const std::vector<double>& b, int k) {   // k-th smallest (0-based)
    double lo = 1e300, hi = -1e300;
    const int n = (int)a.size();
    for (int i = 0; i < n; i++) {
        const double r = (i ? std::fabs(b[i - 1]) : 0.0) + (i < n - 1 ? std::fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
    }
    for (int it = 0; it < 200 && hi - lo > 4e-16 * std::max(std::fabs(lo), std::fabs(hi)); it++) {
        const double mid = 0.5 * (lo + hi);
        if (sturm_count(a, b, mid) > k) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

// ---------------------------------------------------------------------------------- the fit
struct Fit {
    double a0 = 0.0, err = 0.0;
    std::vector<double> res, poles;
};

double eval_fit(const Fit& f, double x) {
    double s = f.a0;
    for (size_t k = 0; k < f.poles.size(); k++) s += f.res[k] / (x + f.poles[k]);
    return s;
}

// poles / residues / constant of the barycentric form with support points z, values f, weights w (first m entries), polished and verified on
// [lo, hi]; returns "" (out filled, out.err = verified error) or the reason this candidate is unusable
std::string extract_fit(double alpha, double l0, double l1, const std::vector<double>& Z, const std::vector<double>& z, const std::vector<double>& f,
                        const std::vector<double>& w, int m, Fit& out) {
    const int M = (int)Z.size();
    // poles: the zeros of D(x) = sum_j w_j / (x - z_j).  For these Stieltjes functions they lie on the negative real axis, where D is smooth
    // (the support points are positive): scan g(s) = sum_j w_j / (s + z_j), s = -x > 0, for sign changes on a fine logarithmic grid, bisect.
    auto g = [&](double s) {
        double d = 0;
        for (int jj = 0; jj < m; jj++) d += w[jj] / (s + z[jj]);
        return d;
    };
    std::vector<double> poles;
    {
        const double e0 = l0 - 12.0 * std::log(10.0), e1 = l1 + 12.0 * std::log(10.0);
        const int steps = (int)((e1 - e0) / std::log(10.0) * 600.0);
        double sp = std::exp(e0), gp = g(sp);
        for (int i = 1; i <= steps; i++) {
            const double sc = std::exp(e0 + (e1 - e0) * i / steps), gc = g(sc);
            if ((gp < 0) != (gc < 0)) {
                double a = std::log(sp), b = std::log(sc);
                const bool neg_a = gp < 0;
                for (int k = 0; k < 100; k++) {
                    const double mid = 0.5 * (a + b);
                    if ((g(std::exp(mid)) < 0) == neg_a) a = mid; else b = mid;
                }
                poles.push_back(std::exp(0.5 * (a + b)));
            }
            sp = sc; gp = gc;
        }
    }
    if ((int)poles.size() != m - 1) return "rational fit produced poles of the wrong kind (off the negative real axis); widen the interval or loosen tol";
    Fit fit;
    fit.poles = poles;
    fit.res.resize(m - 1);
    double swf = 0, sw = 0;
    for (int jj = 0; jj < m; jj++) { swf += w[jj] * f[jj]; sw += w[jj]; }
    fit.a0 = swf / sw;
    bool ok = fit.a0 >= 0;
    for (int k = 0; k < m - 1; k++) {
        double N = 0, Dp = 0;
        for (int jj = 0; jj < m; jj++) {
            const double d = poles[k] + z[jj];
            N += w[jj] * f[jj] / d;
            Dp += w[jj] / (d * d);
        }
        fit.res[k] = N / Dp;
        ok = ok && fit.res[k] > 0;
    }
    if (!ok) return "rational fit produced residues of the wrong sign; widen the interval or loosen tol";
    auto verify = [&](const Fit& c) {          // verification grid, not the fit grid
        const int NV = 1999;
        double err = 0;
        for (int i = 0; i < NV; i++) {
            const double x = std::exp(l0 + (l1 - l0) * i / (NV - 1));
            err = std::max(err, std::fabs(eval_fit(c, x) * std::pow(x, alpha) - 1.0));
        }
        return err;
    };
    fit.err = verify(fit);
    // polish a0 and the residues with the poles fixed (the pole / residue form loses a digit or two against the barycentric one): linear least
    // squares on the relative error over the sample set; kept only if every coefficient stays positive and the verified error does not grow
    {
        const int n = m;
        std::vector<double> A((size_t)M * n), b(M, 1.0), x;
        for (int i = 0; i < M; i++) {
            const double za = std::pow(Z[i], alpha);
            A[i] = za;
            for (int k = 0; k < m - 1; k++) A[(size_t)(k + 1) * M + i] = za / (Z[i] + poles[k]);
        }
        lstsq(A, M, n, b, x);
        bool pos = x[0] >= 0;
        for (int k = 1; k < n; k++) pos = pos && x[k] > 0;
        if (pos) {
            Fit cand = fit;
            cand.a0 = x[0];
            for (int k = 0; k < m - 1; k++) cand.res[k] = x[k + 1];
            cand.err = verify(cand);
            if (cand.err <= fit.err) fit = cand;
        }
    }
    out = fit;
    return "";
}

// x^(-alpha) ~= a0 + sum_k res_k / (x + poles_k) on [lo, hi]; returns "" or the reason the fit was rejected
std::string fit_inverse_power(double alpha, double lo, double hi, double tol, int max_poles, Fit& out) {
    const int M = 3000;
    std::vector<double> Z(M), F(M), Rv(M);
    const double l0 = std::log(lo), l1 = std::log(hi);
    double mean = 0;
    for (int i = 0; i < M; i++) {
        Z[i] = std::exp(l0 + (l1 - l0) * i / (M - 1));
        F[i] = std::pow(Z[i], -alpha);
        mean += F[i];
    }
    mean /= M;
    std::fill(Rv.begin(), Rv.end(), mean);
    std::vector<char> support(M, 0);
    std::vector<double> z, f, w;
    std::vector<std::vector<double>> ws;       // the weight vector of every step: the greedy iteration makes step m a prefix of step m + 1
    const double aaa_tol = 0.1 * tol;
    const int mmax = std::min(max_poles + 1, 64);
    double best = 1e300;
    int since_best = 0;
    for (int it = 0; it < mmax; it++) {
        int j = 0;
        double worst = -1;
        for (int i = 0; i < M; i++) {
            const double e = std::fabs(F[i] - Rv[i]) / std::fabs(F[i]);
            if (e > worst) { worst = e; j = i; }
        }
        if (support[j]) break;                   // the interpolant is exact on the whole sample set
        support[j] = 1;
        z.push_back(Z[j]);
        f.push_back(F[j]);
        const int m = (int)z.size(), rows = M - m;
        std::vector<double> A((size_t)rows * m);     // Loewner matrix on the remaining samples
        for (int jj = 0; jj < m; jj++) {
            int r = 0;
            for (int i = 0; i < M; i++)
                if (!support[i]) A[(size_t)jj * rows + r++] = (F[i] - f[jj]) / (Z[i] - z[jj]);
        }
        min_right_singular_vector(A, rows, m, w);
        ws.push_back(w);
        double err = 0;
        for (int i = 0; i < M; i++) {
            if (support[i]) { Rv[i] = F[i]; continue; }
            double N = 0, D = 0;
            for (int jj = 0; jj < m; jj++) {
                const double c = w[jj] / (Z[i] - z[jj]);
                N += c * f[jj];
                D += c;
            }
            Rv[i] = N / D;
            err = std::max(err, std::fabs(Rv[i] / F[i] - 1.0));
        }
        if (err < aaa_tol) break;
        // at the rounding floor further support points only add pole-zero pairs (Froissart doublets): stop once three steps brought no gain
        if (err < best) { best = err; since_best = 0; }
        else if (++since_best >= 3) break;
    }
    // the last step first; if its poles are spoilt by rounding (or the iteration overshot) the steps before it are complete fits of their own
    std::string first_why;
    Fit best_fit;
    bool have = false;
    for (int m = (int)z.size(); m >= 2; m--) {
        Fit cand;
        std::string why = extract_fit(alpha, l0, l1, Z, z, f, ws[m - 1], m, cand);
        if (!why.empty()) { if (first_why.empty()) first_why = why; continue; }
        if (!have || cand.err < best_fit.err) { best_fit = cand; have = true; }
        if (cand.err <= tol) { out = cand; return ""; }
        if (have && cand.err > 4.0 * best_fit.err) break;       // getting worse: smaller fits will not verify either
    }
    if (!have) return first_why.empty() ? std::string("rational fit: interval too narrow for a fit") : first_why;
    out = best_fit;
    char buf[160];
    snprintf(buf, sizeof buf, "rational fit reached %.2e, requested %.2e", best_fit.err, tol);
    return buf;
}

// wide intervals (small masses) cost digits in double precision: loosen until the fit verifies
std::string fit_loosening(double alpha, double lo, double hi, double tol, Fit& out) {
    for (;;) {
        std::string why = fit_inverse_power(alpha, lo, hi, tol, 40, out);
        if (why.empty()) return why;
        if (tol > 1e-7) return why;
        tol *= 10.0;
    }
}

}  // namespace

// x^(-alpha) ~= a0 + sum_k res[k] / (x + poles[k]) for lam_min <= x <= lam_max, 0 < alpha < 1
extern "C" int lqcd_rational_fit(double alpha, double lam_min, double lam_max, double tol, int max_poles, double* a0, double* res, double* poles,
                                 int* n, double* max_rel_err) {
    ARGCHK(a0 && res && poles && n, "lqcd_rational_fit: null argument");
    ARGCHK(alpha > 0.0 && alpha < 1.0, "lqcd_rational_fit: alpha must lie in (0, 1)");
    ARGCHK(lam_min > 0.0 && lam_min < lam_max, "lqcd_rational_fit: need 0 < lam_min < lam_max");
    ARGCHK(tol > 0.0 && max_poles >= 1, "lqcd_rational_fit: need tol > 0 and max_poles >= 1");
    Fit fit;
    std::string why = fit_inverse_power(alpha, lam_min, lam_max, tol, max_poles, fit);
    if (max_rel_err) *max_rel_err = fit.err;
    if (!why.empty()) { set_error("lqcd_rational_fit: " + why); return LQCD_ERR_NOT_CONVERGED; }
    *n = (int)fit.poles.size();
    *a0 = fit.a0;
    for (int k = 0; k < *n; k++) { res[k] = fit.res[k]; poles[k] = fit.poles[k]; }
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- spectrum of D^+D (Lanczos on the device, scalars on the host)
// |last component| of the normalised eigenvector of the symmetric tridiagonal (a, b) that belongs to its eigenvalue theta: three steps of inverse iteration,
// each a tridiagonal solve with partial pivoting (the elimination of LAPACK's dgtsv: the fill-in is one second superdiagonal).
static double tridiag_last_component(const std::vector<double>& a, const std::vector<double>& b, double theta) {
    const int n = (int)a.size();
    if (n == 1) return 1.0;
    double scale = 0.0;
    for (int i = 0; i < n; i++) scale = std::max(scale, std::fabs(a[i]) + (i < n - 1 ? std::fabs(b[i]) : 0.0));
    const double shift = theta + 1e-13 * std::max(scale, 1e-300);      // off the eigenvalue by a few ulps of the matrix norm: (T - shift) is invertible
    std::vector<double> x(n);
    for (int i = 0; i < n; i++) x[i] = 1.0 + 0.37 * std::sin(1.7 * i + 0.3);      // a start vector that is not orthogonal to anything in particular
    for (int sweep = 0; sweep < 3; sweep++) {
        std::vector<double> d(n), du(n - 1), du2(std::max(0, n - 2), 0.0), dl(n - 1);
        for (int i = 0; i < n; i++) d[i] = a[i] - shift;
        for (int i = 0; i < n - 1; i++) { du[i] = b[i]; dl[i] = b[i]; }
        for (int i = 0; i < n - 1; i++) {
            if (std::fabs(d[i]) >= std::fabs(dl[i])) {          // no row interchange
                const double piv = d[i] != 0.0 ? d[i] : 1e-300 * scale;
                const double m = dl[i] / piv;
                d[i] = piv;
                d[i + 1] -= m * du[i];
                x[i + 1] -= m * x[i];
                if (i < n - 2) du2[i] = 0.0;
            } else {                                            // rows i and i + 1 change places
                const double m = d[i] / dl[i];
                d[i] = dl[i];
                const double t = d[i + 1];
                d[i + 1] = du[i] - m * t;
                du[i] = t;
                if (i < n - 2) { du2[i] = du[i + 1]; du[i + 1] = -m * du2[i]; }
                const double tx = x[i];
                x[i] = x[i + 1];
                x[i + 1] = tx - m * x[i + 1];
            }
        }
        if (d[n - 1] == 0.0) d[n - 1] = 1e-300 * scale;
        x[n - 1] /= d[n - 1];
        if (n > 1) x[n - 2] = (x[n - 2] - du[n - 2] * x[n - 1]) / d[n - 2];
        for (int i = n - 3; i >= 0; i--) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
        double nrm = 0.0;
        for (int i = 0; i < n; i++) nrm = std::max(nrm, std::fabs(x[i]));
        for (int i = 0; i < n; i++) x[i] /= nrm;                // (max-norm first: the solve amplifies by 1 / distance to the spectrum)
        nrm = 0.0;
        for (int i = 0; i < n; i++) nrm += x[i] * x[i];
        nrm = std::sqrt(nrm);
        for (int i = 0; i < n; i++) x[i] /= nrm;
    }
    return std::fabs(x[n - 1]);
}

// Lanczos on D^+D with the residual bound of the extreme Ritz pairs: a Ritz value theta of the k-step tridiagonal with eigenvector s has
// |D^+D y - theta y| = |beta_k s_k| for its Ritz vector y, so an eigenvalue of D^+D lies within that distance of theta.  The smallest Ritz value
// converges from above, slowly for an ill-conditioned operator: the run continues past min_steps (checked every 10 steps) until
// bound_min <= rel_tol theta_min, or gives up at max_steps -- *converged says which.  bound_min / bound_max: the residual bounds of the two ends.
static int lanczos_bounded(lqcd_op_t op, int min_steps, int max_steps, double rel_tol, uint64_t seed, double* theta_min, double* theta_max,
                             double* bound_min, double* bound_max, int* steps_used, bool* converged) {
    lqcd_ctx_s* c = op->ctx;
    ScratchScope sc(c);
    lqcd_spinor_s *v = sc.get(op->kind, LQCD_FULL), *vp = sc.get(op->kind, LQCD_FULL), *w = sc.get(op->kind, LQCD_FULL);
    if (!v || !vp || !w) { set_error("lqcd_estimate_spectrum: out of device memory"); return LQCD_ERR_HIP; }
    LQCHK(lqcd_spinor_gaussian(v, seed));
    double n2 = 0, im = 0;
    LQCHK(lqcd_norm2(v, &n2));
    LQCHK(lqcd_scale(1.0 / std::sqrt(n2), 0.0, v));
    LQCHK(lqcd_spinor_zero(vp));
    std::vector<double> al, be;
    double beta = 0.0;
    *converged = false;
    for (int j = 0; j < max_steps; j++) {
        LQCHK(lqcd_op_apply_DdagD(op, w, v));
        double a = 0;
        LQCHK(lqcd_dot(v, w, &a, &im));
        LQCHK(lqcd_axpy(-a, 0.0, v, w));
        if (j) LQCHK(lqcd_axpy(-beta, 0.0, vp, w));
        al.push_back(a);
        double ww = 0;
        LQCHK(lqcd_norm2(w, &ww));
        beta = std::sqrt(std::max(ww, 0.0));      // beta_{j+1}: the coupling of the (j+1)-step tridiagonal to what it has not seen yet
        const int k = j + 1;
        const bool invariant = beta < 1e-12 * std::fabs(a);
        if (invariant || k == max_steps || (k >= min_steps && (k - min_steps) % 10 == 0)) {
            std::vector<double> bk(be.begin(), be.begin() + (k - 1));
            *theta_min = tridiag_eigenvalue(al, bk, 0);
            *theta_max = tridiag_eigenvalue(al, bk, k - 1);
            *bound_min = invariant ? 0.0 : beta * tridiag_last_component(al, bk, *theta_min);
            *bound_max = invariant ? 0.0 : beta * tridiag_last_component(al, bk, *theta_max);
            *steps_used = k;
            if (*bound_min <= rel_tol * std::fabs(*theta_min)) { *converged = true; break; }
            if (invariant || k == max_steps) break;
        }
        be.push_back(beta);
        LQCHK(lqcd_spinor_copy(vp, v));
        LQCHK(lqcd_spinor_copy(v, w));
        LQCHK(lqcd_scale(1.0 / beta, 0.0, v));
    }
    return LQCD_OK;
}

// the host half of the residual bound on its own (no device): the index-th eigenvalue (ascending) of the symmetric tridiagonal and |last component| of its
// normalised eigenvector, so that the bound can be checked against a dense eigensolver without a GPU (tests/test_rational.py)
extern "C" int lqcd_tridiag_ritz(int n, const double* diag, const double* offdiag, int index, double* theta, double* last_component) {
    ARGCHK(n >= 1 && diag && (offdiag || n == 1) && theta && last_component, "lqcd_tridiag_ritz: bad argument");
    ARGCHK(index >= 0 && index < n, "lqcd_tridiag_ritz: index out of range");
    std::vector<double> a(diag, diag + n), b;
    if (n > 1) b.assign(offdiag, offdiag + (n - 1));
    *theta = tridiag_eigenvalue(a, b, index);
    *last_component = tridiag_last_component(a, b, *theta);
    return LQCD_OK;
}

// the plain k-step estimate (extreme Ritz values, no residual bound): what the bindings' estimate_spectrum returns
extern "C" int lqcd_estimate_spectrum(lqcd_op_t op, int steps, uint64_t seed, double* theta_min, double* theta_max) {
    ARGCHK(op && steps >= 2 && theta_min && theta_max, "lqcd_estimate_spectrum: bad argument");
    double bmin = 0, bmax = 0;
    int used = 0;
    bool conv = false;
    return lanczos_bounded(op, steps, steps, 0.0, seed, theta_min, theta_max, &bmin, &bmax, &used, &conv);
}

// ---------------------------------------------------------------------------------- FermiAction handle
struct lqcd_action_s {
    lqcd_op_s* op = nullptr;
    double nf = 2, alpha = 1, eps = 1e-19;
    int maxiter = 3000;
    bool rational = false, evensite = false, explicit_interval = false;
    int lanczos_steps = 60, refits = 0;
    int lanczos_used = 0;              // steps the last Lanczos run took, the residual bound |beta_k s_k| of its smallest Ritz value
    double ritz_bound = 0.0;
    double tol_action = 1e-12, tol_md = 1e-8;
    double lo = 0, hi = 0;
    Fit fit[3];       // 0: x^(-alpha) for the action, 1: the same for the MD force (looser), 2: x^(alpha/2 - 1) for the heat bath
};

// Spectral interval of D^+D on the current links for the Wilson(-clover) rational action, CERTIFIED: the Lanczos run continues (up to 8 x
// rhmc_lanczos_steps) until the residual bound of the smallest Ritz value is below 10 % of it; *tmin / *tmax are the Ritz values moved outwards by their
// bounds.  need_lo: the caller has no lower edge of its own -- an unconverged smallest Ritz value is then an error (a fit made from it could be used
// outside its interval without anybody noticing), to be resolved with rhmc_lambda_min or more rhmc_lanczos_steps.
static int action_spectrum(lqcd_action_s* fa, double* tmin, double* tmax, bool need_lo) {
    double th0 = 0, th1 = 0, b0 = 0, b1 = 0;
    bool conv = false;
    const int min_steps = std::max(2, fa->lanczos_steps);
    LQCHK(lanczos_bounded(fa->op, min_steps, 8 * min_steps, 0.1, 4711, &th0, &th1, &b0, &b1, &fa->lanczos_used, &conv));
    fa->ritz_bound = b0;
    if (!conv && need_lo) {
        char buf[400];
        snprintf(buf, sizeof buf, "FermiAction: the smallest Ritz value of D'D, %.3e, has not converged after %d Lanczos steps (residual bound %.3e): "
                 "the residual bound of the smallest Ritz value did not converge: no lower edge for the rational fit -- pass rhmc_lambda_min (and rhmc_lambda_max), or raise rhmc_lanczos_steps", th0, fa->lanczos_used, b0);
        set_error(buf);
        return LQCD_ERR_NOT_CONVERGED;
    }
    *tmin = std::max(th0 - b0, 0.0);
    *tmax = th1 + b1;
    return LQCD_OK;
}

static int action_fit(lqcd_action_s* fa, double lo, double hi) {
    const double al[3] = {fa->alpha, fa->alpha, 1.0 - 0.5 * fa->alpha};
    const double tol[3] = {fa->tol_action, fa->tol_md, fa->tol_action};
    Fit f[3];
    for (int i = 0; i < 3; i++) {
        std::string why = fit_loosening(al[i], lo, hi, tol[i], f[i]);
        if (!why.empty()) { set_error("FermiAction: " + why); return LQCD_ERR_NOT_CONVERGED; }
    }
    for (int i = 0; i < 3; i++) fa->fit[i] = f[i];
    fa->lo = lo; fa->hi = hi;
    return LQCD_OK;
}

extern "C" int lqcd_action_create(lqcd_op_t op, double nf, double eps, int maxiter, int nparams, const char* const* keys, const double* values,
                                  lqcd_action_t* out) {
    ARGCHK(op && out, "lqcd_action_create: null argument");
    ARGCHK(nparams == 0 || (keys && values), "lqcd_action_create: parameter arrays missing");
    lqcd_action_s* fa = new lqcd_action_s;
    fa->op = op;
    fa->eps = eps;
    fa->maxiter = maxiter;
    const int kind = op->kind;
    if (nf <= 0) nf = kind == LQCD_STAGGERED ? 4 : 2;    // the reference's defaults (universe.jl:106-110 passes p.Nf)
    fa->nf = nf;
    bool force_rational = false, have_lo = false, have_hi = false;
    double plo = 0, phi = 0;
    for (int i = 0; i < nparams; i++) {
        const std::string k = keys[i] ? keys[i] : "";
        if (k == "force_rational") force_rational = values[i] != 0;
        else if (k == "rhmc_lambda_min") { plo = values[i]; have_lo = true; }
        else if (k == "rhmc_lambda_max") { phi = values[i]; have_hi = true; }
        else if (k == "rhmc_tol_action") fa->tol_action = values[i];
        else if (k == "rhmc_tol_MD") fa->tol_md = values[i];
        else if (k == "rhmc_lanczos_steps") fa->lanczos_steps = (int)values[i];
        else { delete fa; set_error("lqcd_action_create: unknown parameter " + k); return LQCD_ERR_ARG; }
    }
    // D^+D carries 2 Wilson flavours / 8 staggered tastes (4 when the pseudofermion lives on the even sites): anything else is
    // S_f = eta^+ (D^+D)^(-Nf/n0) eta
    if (kind == LQCD_DOMAINWALL) {      // two flavours with the Pauli-Villars field (domainwall.hip); nothing to fit
        if (nf != 2 || force_rational) { delete fa; set_error("FermiAction: the Domainwall action is the two-flavour one (Nf = 2)"); return LQCD_ERR_UNSUPPORTED; }
        *out = fa;
        return LQCD_OK;
    }
    const double n0 = kind == LQCD_WILSON ? 2 : 8;
    fa->evensite = kind == LQCD_STAGGERED && nf == 4;
    fa->rational = (kind == LQCD_WILSON && nf != 2) || (kind == LQCD_STAGGERED && nf != 4 && nf != 8) || force_rational;
    if (fa->rational) {
        fa->evensite = false;
        if (!(nf > 0 && nf < n0)) {
            delete fa;
            set_error("FermiAction: Nf = " + std::to_string(nf) + " outside (0, " + std::to_string((int)n0) + ") for this operator");
            return LQCD_ERR_UNSUPPORTED;
        }
        fa->alpha = nf / n0;
        fa->explicit_interval = have_lo && have_hi;
        double lo, hi;
        if (fa->explicit_interval) { lo = plo; hi = phi; }
        else if (kind == LQCD_STAGGERED) {     // D^+D = m^2 - D_hop^2 with |D_hop| <= 4
            lo = op->km * op->km * (1.0 - 1e-9);
            hi = (op->km * op->km + 16.0) * (1.0 + 1e-9);
        } else {                               // Wilson(-clover): no analytic lower bound -- Lanczos estimate on the current links with a margin
            double tmin = 0, tmax = 0;
            int st = action_spectrum(fa, &tmin, &tmax, !have_lo);
            if (st != LQCD_OK) { delete fa; return st; }
            lo = have_lo ? plo : 0.5 * tmin;      // tmin: smallest Ritz value minus its residual bound |beta_k s_k| -- SOME eigenvalue lies that close to it; that none lies below is not proven (no
                                                  // reorthogonalisation: ghosts), which is why every use of the interval is guarded by lqcd_action_check_interval
            hi = have_hi ? phi : 1.2 * tmax;
        }
        if (!(lo > 0 && lo < hi)) { delete fa; set_error("FermiAction: need 0 < rhmc_lambda_min < rhmc_lambda_max"); return LQCD_ERR_ARG; }
        int st = action_fit(fa, lo, hi);
        if (st != LQCD_OK) { delete fa; return st; }
    }
    *out = fa;
    return LQCD_OK;
}

extern "C" int lqcd_action_destroy(lqcd_action_t fa) {
    delete fa;
    return LQCD_OK;
}

extern "C" int lqcd_action_set_solver(lqcd_action_t fa, double eps, int maxiter) {
    ARGCHK(fa && maxiter >= 1, "lqcd_action_set_solver: bad argument");
    fa->eps = eps;
    fa->maxiter = maxiter;
    return LQCD_OK;
}

extern "C" int lqcd_action_get(lqcd_action_t fa, const char* key, double* value) {
    ARGCHK(fa && key && value, "lqcd_action_get: null argument");
    const std::string k = key;
    if (k == "rational") *value = fa->rational;
    else if (k == "evensite") *value = fa->evensite;
    else if (k == "Nf") *value = fa->nf;
    else if (k == "alpha") *value = fa->alpha;
    else if (k == "lambda_min") *value = fa->lo;
    else if (k == "lambda_max") *value = fa->hi;
    else if (k == "interval_refits") *value = fa->refits;
    else if (k == "lanczos_steps_used") *value = fa->lanczos_used;
    else if (k == "ritz_bound") *value = fa->ritz_bound;
    else if (k == "explicit_interval") *value = fa->explicit_interval;
    else { set_error("lqcd_action_get: unknown key " + k); return LQCD_ERR_ARG; }
    return LQCD_OK;
}

extern "C" int lqcd_action_coefficients(lqcd_action_t fa, int which, double* a0, double* res, double* poles, int capacity, int* n, double* max_rel_err) {
    ARGCHK(fa && fa->rational && which >= 0 && which < 3 && n, "lqcd_action_coefficients: not a rational action / bad selector");
    const Fit& f = fa->fit[which];
    *n = (int)f.poles.size();
    if (a0) *a0 = f.a0;
    if (max_rel_err) *max_rel_err = f.err;
    if (res && poles) {
        ARGCHK(capacity >= *n, "lqcd_action_coefficients: arrays too short");
        for (int k = 0; k < *n; k++) { res[k] = f.res[k]; poles[k] = f.poles[k]; }
    }
    return LQCD_OK;
}

extern "C" int lqcd_action_set_coefficients(lqcd_action_t fa, int which, double a0, int n, const double* res, const double* poles) {
    ARGCHK(fa && fa->rational && which >= 0 && which < 3 && n >= 1 && res && poles, "lqcd_action_set_coefficients: bad argument");
    Fit f;
    f.a0 = a0;
    f.res.assign(res, res + n);
    f.poles.assign(poles, poles + n);
    for (int k = 0; k < n; k++) ARGCHK(poles[k] >= 0, "lqcd_action_set_coefficients: poles must be >= 0");
    fa->fit[which] = f;
    return LQCD_OK;
}

// Wilson(-clover) rational action: the spectrum of D^+D has no analytic bound and drifts along the HMC stream, and partial fractions used outside
// their fit interval silently bias S_f, the heat bath and the force.  Run at the heat bath and at every evaluation of the action: a Lanczos run on
// the CURRENT links.  An interval that came from the estimate (margins 0.5 / 1.2 at the fit; the smallest Ritz value converges from above) is
// refitted as soon as its lower edge exceeds 0.6 x the smallest Ritz value or its upper edge falls below 1.1 x the largest; an interval the
// caller fixed raises once a Ritz value lies outside it.
extern "C" int lqcd_action_check_interval(lqcd_action_t fa) {
    ARGCHK(fa, "lqcd_action_check_interval: null");
    if (!(fa->rational && fa->op->kind == LQCD_WILSON)) return LQCD_OK;
    double tmin = 0, tmax = 0;
    LQCHK(action_spectrum(fa, &tmin, &tmax, !fa->explicit_interval));      // ends: Ritz values moved outwards by their residual bounds (estimates, not certificates)
    if (fa->explicit_interval) {
        if (fa->lo <= tmin && tmax <= fa->hi) return LQCD_OK;
        char buf[320];
        snprintf(buf, sizeof buf, "FermiAction: the spectrum of D'D on the current links, Ritz values [%.3e, %.3e], is not inside the fit interval "
                 "[%.3e, %.3e] given by rhmc_lambda_min / rhmc_lambda_max", tmin, tmax, fa->lo, fa->hi);
        set_error(buf);
        return LQCD_ERR_ARG;
    }
    if (fa->lo <= 0.6 * tmin && fa->hi >= 1.1 * tmax) return LQCD_OK;
    LQCHK(action_fit(fa, std::min(fa->lo, 0.4 * tmin), std::max(fa->hi, 1.25 * tmax)));
    fa->refits++;
    return LQCD_OK;
}

static int action_bind(lqcd_action_s* fa, lqcd_gauge_s* U) {      // the fa.D(U) of the callers
    if (U && U != fa->op->gauge) return lqcd_op_set_gauge(fa->op, U);
    return LQCD_OK;
}

// gauss_sampling_in_action!(xi, U, fa) (src/md/standardMD.jl:95): xi distributed as exp(-xi^+ xi), i.e. re and im of variance 1/2
extern "C" int lqcd_action_gauss_sampling(lqcd_action_t fa, lqcd_spinor_t xi, uint64_t seed) {
    ARGCHK(fa && xi, "lqcd_action_gauss_sampling: null argument");
    LQCHK(lqcd_spinor_gaussian(xi, seed));
    return lqcd_scale(std::sqrt(0.5), 0.0, xi);
}

// sample_pseudofermions!(eta, U, fa, xi) (standardMD.jl:96): eta = D^+ xi (restricted to the even sites for 4 staggered tastes); rational action:
// eta = (D^+D)^(alpha/2) xi = D^+D r_sampling(D^+D) xi so that eta^+ (D^+D)^(-alpha) eta = xi^+ xi
extern "C" int lqcd_action_sample_pseudofermions(lqcd_action_t fa, lqcd_gauge_t U, lqcd_spinor_t eta, lqcd_spinor_t xi) {
    ARGCHK(fa && eta && xi && eta != xi, "lqcd_action_sample_pseudofermions: need distinct fields");
    LQCHK(action_bind(fa, U));
    lqcd_op_s* op = fa->op;
    if (op->kind == LQCD_DOMAINWALL) return dw_sample(op, eta, xi, fa->eps, fa->maxiter);
    LQCHK(check_full(op, eta, xi, "lqcd_action_sample_pseudofermions"));
    if (fa->rational) {
        LQCHK(lqcd_action_check_interval(fa));
        ScratchScope sc(op->ctx);
        lqcd_spinor_s* t = sc.get(op->kind, LQCD_FULL);
        if (!t) { set_error("sample_pseudofermions: out of device memory"); return LQCD_ERR_HIP; }
        const Fit& f = fa->fit[2];
        LQCHK(lqcd_rational_apply(op, t, xi, f.a0, (int)f.poles.size(), f.res.data(), f.poles.data(), fa->eps, fa->maxiter, nullptr));
        return lqcd_op_apply_DdagD(op, eta, t);
    }
    LQCHK(lqcd_op_apply(op, eta, xi, 1));
    if (fa->evensite) {
        lqcd_ctx_s* c = op->ctx;
        HIPCHK(hipMemsetAsync(eta->data + eta->elems / 2, 0, (eta->elems / 2) * sizeof(double2), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return LQCD_OK;
}

// evaluate_FermiAction(fa, U, eta) (src/updates/standardHMC.jl:69,71).  X / Y (optional) receive (D^+D)^-1 eta / D X of the exact actions, X the
// rational image r(D^+D) eta otherwise
extern "C" int lqcd_action_evaluate(lqcd_action_t fa, lqcd_gauge_t U, lqcd_spinor_t eta, lqcd_spinor_t X, lqcd_spinor_t Y, double* Sf, int* iters) {
    ARGCHK(fa && eta, "lqcd_action_evaluate: null argument");
    LQCHK(action_bind(fa, U));
    lqcd_op_s* op = fa->op;
    if (op->kind == LQCD_DOMAINWALL) return dw_action(op, eta, X, Y, fa->eps, fa->maxiter, Sf, iters);
    ScratchScope sc(op->ctx);
    if (!X) X = sc.get(op->kind, LQCD_FULL);
    if (!X) { set_error("evaluate_FermiAction: out of device memory"); return LQCD_ERR_HIP; }
    if (fa->rational) {
        LQCHK(lqcd_action_check_interval(fa));
        const Fit& f = fa->fit[0];
        LQCHK(lqcd_rational_apply(op, X, eta, f.a0, (int)f.poles.size(), f.res.data(), f.poles.data(), fa->eps, fa->maxiter, iters));
        double re = 0, im = 0;
        LQCHK(lqcd_dot(eta, X, &re, &im));
        if (Sf) *Sf = re;
        return LQCD_OK;
    }
    if (!Y) Y = sc.get(op->kind, LQCD_FULL);
    if (!Y) { set_error("evaluate_FermiAction: out of device memory"); return LQCD_ERR_HIP; }
    return lqcd_fermi_action(op, eta, X, Y, fa->eps, fa->maxiter, Sf, iters);
}

// calc_UdSfdU!(UdSfdU, fa, U, eta) (src/md/AbstractMD.jl:129): out = G, the force field in the convention of lqcd_fermion_force
extern "C" int lqcd_action_force(lqcd_action_t fa, lqcd_gauge_t U, lqcd_gauge_t out, lqcd_spinor_t eta, double* Sf, int* iters) {
    ARGCHK(fa && out && eta, "lqcd_action_force: null argument");
    LQCHK(action_bind(fa, U));
    lqcd_op_s* op = fa->op;
    if (op->kind == LQCD_DOMAINWALL) return dw_force(op, out, eta, fa->eps, fa->maxiter, Sf, iters);
    if (fa->rational) {
        const Fit& f = fa->fit[1];
        if (Sf) *Sf = 0.0;
        return lqcd_rational_force(op, out, eta, (int)f.poles.size(), f.res.data(), f.poles.data(), fa->eps, fa->maxiter, iters);
    }
    return lqcd_calc_UdSfdU(op, out, eta, fa->eps, fa->maxiter, Sf, iters);
}
