/*
 * lqcd_oracle.h -- CPU oracle for the Dirac-solver hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  Nothing under latticeqcd.jl_amd/ (the product) may import, link or call it.
 *
 * PARITY STATUS: the arithmetic of this path lives in the un-vendored Julia packages
 * LatticeDiracOperators.jl (compat 0.6.1, /root/reference/Project.toml:11,27) and
 * Gaugefields.jl (compat 0.4-0.7, Project.toml:8,24); neither their source nor a Julia
 * runtime exists in the build container.  This file restates their published algorithm
 *  and is pinned against the reference ONLY where the reference
 * holds data: gauge-configuration formats, site/link index order and plaquettes of the
 * fixtures under /root/reference/test/confs_* (tests/golden/), and -- end to end -- the
 * reference's own HMC test: tests/test_oracle_md.py repeats test/runtests.jl:88-99 with
 * test/test_wilson.toml (thermalised 4^4 start, beta 5.7, kappa 0.141139, dtau 0.05, 20 MD
 * steps, Sexton-Weingarten N = 10, 10 trajectories) through this oracle's Dslash, CG, fermion
 * force, gauge force and integrator and meets the reference's criterion (final plaquette
 * within 10 % of test/debugplaqdata.txt:7).  That criterion is statistical and loose (it does
 * reject a pseudofermion weight of exp(-S_f/2), tests/test_gpu_md.py), so at the level of a
 * single Dslash application or CG solve the status remains **parity unpinned**: no
 * reference data exists there.  The operator conventions are defended by
 * convention-independent identities in tests/test_oracle_identities.py and test_oracle_md.py,
 * and (round 4) the conventions that are OBSERVABLE are pinned to published physics through the
 * HIP path this oracle checks: the quenched SU(3) Wilson plaquette at beta 5.7 / 6.0 and the
 * quenched Wilson pion mass at beta 5.7, kappa 0.1600 / 0.1650 (Butler et al., Nucl. Phys. B 430
 * (1994) 179) come out within 1 %, the quenched staggered Goldstone pion at beta 6.0, m = 0.01 / 0.03
 * (Gupta et al., Phys. Rev. D 43 (1991) 2003) within 2 % with m_pi^2 proportional to m
 * and the clover term reproduces kappa_c(beta 6.0, c_sw 1.769) = 0.135196 to 1.6e-4
 * (tests/test_gpu_quenched_literature.py) -- the normalisation of beta, of kappa and of the staggered
 * mass term, r = 1, the hop structure and the staggered phases.  That is not parity with the
 * reference's own bits.
 *
 * Memory layouts are the reference's host layouts (Julia column-major):
 *   gauge  U[mu][a,b,ix,iy,iz,it]  (src/updates/givenconfigurations.jl:49)
 *          -> flat index  a + 3*(b + 3*(site + V*mu)),  site = ix + NX*(iy + NY*(iz + NZ*it))
 *   Wilson psi[ic,ix,iy,iz,it,is]  (src/measurements/unusedfiles/measure_Pion_correlator.jl:244,376)
 *          -> flat index  ic + 3*(site + V*is)
 *   staggered psi[ic,ix,iy,iz,it,1] -> ic + 3*site
 * All arrays are interleaved (re,im) doubles.
 */
#ifndef LQCD_ORACLE_H
#define LQCD_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_WILSON = 0, ORC_STAGGERED = 1 };

/* number of threads used by the site loops (1 = the reference's serial loop) */
void orc_set_threads(int n);
/* dst <- src, every thread copying (= first touching) the index ranges it owns in the threaded loops: nblk blocks of n / nblk complex numbers (bench.py, all-cores leg) */
void orc_numa_copy(double* dst, const double* src, long n, int nblk);
int orc_get_threads(void);

/* plaquette, normalisation 1/(6*V*NC): src/measurements/unusedfiles/measure_plaquette.jl:41 */
double orc_plaquette(const double* U, const int L[4]);
/* max_{links} ||U U^dagger - 1||_max */
double orc_unitarity_dev(const double* U, const int L[4]);

/* y = D x or D^dagger x.  Wilson: D = 1 - kappa sum_nu[(r-g_nu)U_nu(n)d(n+nu) + (r+g_nu)U^+_nu(n-nu)d(n-nu)]
 * (parameters from src/system/universe.jl:111-116,132-135) */
void orc_wilson_D(double* out, const double* U, const double* in, const int L[4], double kappa,
                  double r, const int bc[4], int dagger);
/* y = (m + 1/2 sum_nu eta_nu(n)[U_nu(n)x(n+nu) - U^+_nu(n-nu)x(n-nu)]) x   (universe.jl:106-110) */
void orc_staggered_D(double* out, const double* U, const double* in, const int L[4], double mass,
                     const int bc[4], int dagger);
/* only the hopping part restricted to output sites of one parity (0 even, 1 odd); other sites zeroed.
 * Wilson: out = sum_nu[(r-g)U x+ + (r+g)U^+ x-]  (no kappa);  dagger flips the gamma signs. */
void orc_wilson_hop_parity(double* out, const double* U, const double* in, const int L[4], double r,
                           const int bc[4], int dagger, int out_parity);

/* BLAS-1 on n complex numbers */
void orc_dot(const double* a, const double* b, long n, double* re, double* im); /* sum conj(a) b */
void orc_axpy(double ar, double ai, const double* x, double* y, long n);       /* y += a x */

/* Krylov solvers; stopping rule real(r.r) < eps (absolute, squared) -- SURVEY.md 3.3;
 * defaults eps=1e-19, maxiter=3000 from src/system/parameter_structs.jl:174-175.
 * x holds the initial guess on entry.  Return 0 = converged, 1 = not converged. */
int orc_cg_DdagD(int kind, double* x, const double* U, const double* b, const int L[4], double kappa_or_mass,
                 double r, const int bc[4], double eps, int maxiter, int* iters, double* final_rr);
int orc_bicg(int kind, double* x, const double* U, const double* b, const int L[4], double kappa_or_mass, double r, const int bc[4],
             int dagger, double eps, int maxiter, int* iters, double* final_rr);
int orc_bicgstab(int kind, double* x, const double* U, const double* b, const int L[4], double kappa_or_mass,
                 double r, const int bc[4], int dagger, double eps, int maxiter, int* iters, double* final_rr);
/* even-odd (Schur) preconditioned BiCGStab for Wilson D x = b (full-lattice in/out) */
int orc_wilson_bicgstab_eo(double* x, const double* U, const double* b, const int L[4], double kappa, double r,
                           const int bc[4], int dagger, double eps, int maxiter, int* iters, double* final_rr);

/* multi-shift CG: (D^+D + sigma_j) x_j = b, j < ns, plus the unshifted solution x0 (RHMC solver).
 * xs holds ns vectors back to back.  Stops when rr * max(1, max_j zeta_j^2) < eps. */
int orc_multishift_cg(int kind, double* x0, double* xs, const double* U, const double* b, const int L[4], double kappa_or_mass,
                      double r, const int bc[4], const double* sigma, int ns, double eps, int maxiter, int* iters, double* final_rr);

/* pseudofermion action S_f = eta^+ (D^+D)^-1 eta (evaluate_FermiAction, src/updates/standardHMC.jl:71) by CG from a zero
 * guess; also returns X = (D^+D)^-1 eta and, if Y != NULL, Y = D X.  *status: 0 converged, 1 not. */
double orc_fermi_action(int kind, double* X, double* Y, const double* U, const double* eta, const int L[4], double kappa_or_mass,
                        double r, const int bc[4], double eps, int maxiter, int* iters, int* status);
/* fermion force G_mu(n) = "U dS_f/dU" (calc_UdSfdU!, src/md/AbstractMD.jl:129), gauge layout, defined by
 *   d/d eps S_f[U_mu(n) -> exp(i eps T) U_mu(n)] = -2 Im tr(T G_mu(n))  for Hermitian T   (formulas: lqcd_oracle.c) */
void orc_wilson_force(double* G, const double* U, const double* X, const double* Y, const int L[4], double kappa, double r,
                      const int bc[4]);
void orc_staggered_force(double* G, const double* U, const double* X, const double* Y, const int L[4], const int bc[4]);

/* gauge side of the MD step (P_update!/U_update!, src/md/AbstractMD.jl:78-118; conventions: lqcd_oracle.c).
 * Momenta P are traceless anti-Hermitian 3x3 matrices in the gauge layout; K = -sum tr P^2; dU/dtau = P U;
 * S_g = -(beta/3) sum_plaq Re tr U_p; force fields G obey dS/d eps[U -> exp(i eps T)U] = -2 Im tr(T G); dP/dtau = TA(G). */
double orc_gauge_action(const double* U, const int L[4], double beta);
void orc_gauge_force(double* G, const double* U, const int L[4], double beta);      /* G = -(beta/6) U * staples */
void orc_momentum_add_ta(double* P, double c, const double* G, const int L[4]);     /* P += c TA(G) */
double orc_momentum_action(const double* P, const int L[4]);                        /* K = -sum tr P^2 */
void orc_link_update(double* U, const double* P, double dt, const int L[4]);        /* U <- exp(dt P) U */

/* clover term (no reference behaviour exists -- textbook definition, see lqcd_oracle.c):
 * clov[V][12][12] = A(x) = 1 + i kappa c_sw sum_{mu<nu} sigma_{mu nu} F_{mu nu}(x), row-major in (s*3+c) */
void orc_clover_build(double* clov, const double* U, const int L[4], double kappa, double csw);
void orc_wilson_clover_D(double* out, const double* U, const double* clov, const double* in, const int L[4], double kappa,
                         double r, const int bc[4], int dagger);
int orc_cg_clover(double* x, const double* U, const double* clov, const double* b, const int L[4], double kappa, double r,
                  const int bc[4], double eps, int maxiter, int* iters, double* final_rr);

/* A^-1 site by site, and the even-odd (Schur) preconditioned BiCGStab for D_sw (or D_sw^+) built on it */
void orc_clover_invert(double* inv, const double* clov, const int L[4]);
int orc_wilson_clover_bicgstab_eo(double* x, const double* U, const double* clov, const double* b, const int L[4], double kappa, double r,
                                  const int bc[4], int dagger, double eps, int maxiter, int* iters, double* final_rr);

/* "U dS/dU" of the clover term for S_f = phi^+ (D_sw^+ D_sw)^-1 phi (X = (D_sw^+ D_sw)^-1 phi, Y = D_sw X); add orc_wilson_force for the
 * hopping part.  Same convention as the other force fields. */
void orc_clover_force(double* G, const double* U, const double* X, const double* Y, const int L[4], double kappa, double csw, int accumulate);

/* fixed-length CG window with the exit test disabled (timing only): runs exactly niter iterations */
void orc_cg_DdagD_fixed(int kind, double* x, const double* U, const double* b, const int L[4],
                        double kappa_or_mass, double r, const int bc[4], int niter);

#ifdef __cplusplus
}
#endif
#endif
