/*
 * lqcd_hip.h -- C ABI of liblqcd_hip.so, the MI355X-native Dirac-solver hot path for LatticeQCD.jl.
 *
 * The reference has no FFI for this path: the seam is Julia multiple dispatch on types owned by
 * LatticeDiracOperators.jl / Gaugefields.jl.  Every entry point below cites the
 * reference call site / generic function whose method a `ccall` stub replaces (INTEGRATION.md shows
 * the Julia binding).  Citations are relative to /root/reference.
 *
 * Conventions
 *   - every function returns int: LQCD_OK (0) or an LQCD_ERR_* code; lqcd_last_error() gives the message
 *     (the Julia side turns non-zero into `error(msg)`, mirroring the reference's error(...) strings);
 *   - host buffers are caller-owned and only borrowed for the call; device memory is library-owned
 *     behind opaque handles;
 *   - all calls are synchronous at return unless named *_async;
 *   - host layouts are the reference's (Julia column-major), interleaved (re,im) doubles, LOCAL sub-lattice
 *     of the calling rank, no wing:
 *       gauge     U[mu][a,b,ix,iy,iz,it]   -> a + 3*(b + 3*(site + V*mu))     (src/updates/givenconfigurations.jl:49)
 *       Wilson    psi[ic,ix,iy,iz,it,is]   -> ic + 3*(site + V*is)             (src/measurements/unusedfiles/measure_Pion_correlator.jl:244,376)
 *       staggered psi[ic,ix,iy,iz,it,1]    -> ic + 3*site
 *     with site = ix + NX*(iy + NY*(iz + NZ*it)), all 0-based.
 *   - arithmetic is fp64 throughout.
 */
#ifndef LQCD_HIP_H
#define LQCD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lqcd_ctx_s* lqcd_ctx_t;
typedef struct lqcd_gauge_s* lqcd_gauge_t;
typedef struct lqcd_spinor_s* lqcd_spinor_t;
typedef struct lqcd_op_s* lqcd_op_t;
typedef struct lqcd_action_s* lqcd_action_t;

enum {
    LQCD_OK = 0,
    LQCD_ERR_ARG = 1,           /* bad argument / shape mismatch */
    LQCD_ERR_HIP = 2,           /* HIP runtime error (no device, OOM, launch failure) */
    LQCD_ERR_NOT_CONVERGED = 3, /* solver hit maxiter: the reference raises error(...) here */
    LQCD_ERR_COMM = 4,          /* RCCL error, or a peer-mapped exchange that gave up waiting (dead rank) */
    LQCD_ERR_UNSUPPORTED = 5
};

enum { LQCD_WILSON = 0, LQCD_STAGGERED = 1,          /* "Dirac_operator" => "Wilson" | "Staggered"  (src/system/universe.jl:103-116) */
       LQCD_DOMAINWALL = 2 };                        /* ... | "Domainwall" (universe.jl:116-128): lqcd_op_create_domainwall, five-dimensional fields */
enum { LQCD_FULL = 0, LQCD_EVEN = 1, LQCD_ODD = 2 }; /* site subset held by a spinor */
enum { LQCD_LAYOUT_REFERENCE = 0, /* memory image of the Julia arrays, see above */
       LQCD_LAYOUT_DISK = 1 };    /* ILDG / BridgeText flat order t,z,y,x,mu,a,b */

/* ---------------------------------------------------------------- host-only helpers (no GPU needed) */
int lqcd_version(void);
const char* lqcd_last_error(void);
/* number of HIP devices visible (0 if none) */
int lqcd_device_count(void);
/* free / total HBM of a device in bytes (diagnostics; the leak test of tests/test_gpu_lifecycle.py) */
int lqcd_device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes);
/* reference site linearisation ix + NX*(iy + NY*(iz + NZ*it)) */
int64_t lqcd_index_lex(const int L[4], int x, int y, int z, int t);
/* device (checkerboard) position of a site: parity = (x+y+z+t)&1, cb = (x>>1) + (NX/2)*(y + NY*(z + NZ*t)) */
int lqcd_index_cb(const int L[4], int x, int y, int z, int t, int* parity, int64_t* cb);
/* inverse of lqcd_index_cb */
int lqcd_coords_cb(const int L[4], int parity, int64_t cb, int xyzt[4]);
/* local extents, origin and neighbour ranks of `rank` in the PE grid (the reference's PEs concept, src/mpirun.jl:17-19;
 * rank = px + PX*(py + PY*(pz + PZ*pt))) */
int lqcd_decompose(const int global_L[4], const int pe_grid[4], int rank, int local_L[4], int origin[4],
                   int rank_fwd[4], int rank_bwd[4]);

/* ---------------------------------------------------------------- context */
/* One context = one rank's sub-lattice on one GPU.  pe_grid = {1,1,1,1}, rank 0 for a single GPU. */
int lqcd_ctx_create(lqcd_ctx_t* ctx, int device, const int global_L[4], const int pe_grid[4], int rank);
int lqcd_ctx_destroy(lqcd_ctx_t ctx);
int lqcd_ctx_sync(lqcd_ctx_t ctx);
/* tuning knobs (kernel variants); unknown key -> LQCD_ERR_ARG.  Keys: dslash_variant (0 site-per-lane, 1 direction split
 * [default], 2 hop split, 3 persistent hop split, 4 lane split = four directions in the four 16-lane rows of a wave combined by
 * v_permlane swaps, 5 direction split with the footprint of four workgroups per CU, 6 direction split with the x / y neighbour spinors staged
 * through LDS, 7 both parities of a chunk in one 512-thread workgroup; variants 2-8 exist only in LQCD_VARIANTS=1 builds of the library --
 * read-only key variants_built -- and run variant 1 otherwise), dslash_block, xcd_remap, xcd_nsub, xcd_ysplit, cg_fused (0 reference form, 1, 2 fused
 * [default]), graph (1: hipGraph replay of CG bursts), gauge_recon (12 [default]: the split kernels read two rows per link and rebuild the
 * third -- applied only while every link of the field is unitary to 1e-14, results within the fp64 Dslash tolerance; 18: all
 * 18 stored reals are always read), recon_active (read-only: did the last Wilson application use the 12-real links), nt_gauge (bit 0 [default]: the backward = last use of a link is a non-temporal load; bit 1: the forward use too), nt_store (1 [default]:
 * non-temporal output stores), lds_pad_kb, persist_per_cu, dslash_pipe (forms of the Wilson r = 1 direction-split kernel on lattices whose z-planes are whole chunks;
 * all bit-identical: 2 [default]: scalar (wave-uniform) addressing, one workgroup per chunk, 12-real fp64 links only; 1: persistent workgroups with the next chunk's
 * loads in flight across the barrier, per-XCD in-order queue -- pipe_per_cu / pipe_grid / pipe_min_chunks shape its grid; 3: pipe_chunks_per_wg consecutive chunks per
 * workgroup, pipelined; 0: plain variant 1.  Forms 1 and 3 measured slower, LABNOTES.md section 2 "Round 3"), mixed_pair32 (1 [default]: the mixed-precision Wilson
 * solvers use the fp32 site-pair kernel on unpartitioned lattices with T % 4 == 0; read-only pair32_active), mixed_links16 (that kernel reads its links as int16 fixed point;
 * 1 [default]: in the even-odd BiCGStab of the action / force solves, 2: in every mixed-precision solver, 0: fp32 links), bicg_reliable (1: the fp32 chain of that BiCGStab goes on
 * behind a correction step instead of starting again; default 0), dw_fused_cg (1 [default]: Domainwall solves run the fused CG iteration on the five-dimensional launch), clover_hop_s (1 [default]: the hops of the even-odd Wilson-clover
 * solver run the scalar-addressing kernel with the inverse clover blocks in its epilogue), stag_both (1: the staggered split kernel issues the loads of both hops back to back);
 * solvers / actions: mixed_action_solver (1: lqcd_fermi_action / lqcd_calc_UdSfdU / the staggered rational entries solve with the
 * mixed-precision CG; 2: the same, and every rational entry solves all its poles with lqcd_solve_multishift_mixed_cg -- measured slower than the fp64
 * multi-shift CG at 48^3x96 with 18 poles, default 0), staggered_parity_solve (1 [default]: half-lattice CG for a staggered eta whose odd half is zero),
 * md_remap (1 [default]: the staple sweep follows the stencil's XCD-aware workgroup map), md_reunitarize (1 [default]: lqcd_gauge_exp_update projects the updated
 * links back onto SU(3) in the same pass; 0: the reference's literal update), cg_fold_scalars (1 [default]: several ranks, the scalar steps behind the two all-reduces of a
 * CG iteration run in the consumers' prologues), nt_blas (1 [default]: non-temporal loads / stores in the CG update kernels),
 * cg_skip_done, cg_defer_x (2: the fused CG updates x every second iteration with both search directions, p alternating between
 * two buffers -- 9 instead of 10 spinor passes per iteration on average, identical iterates; K = 3..8 (round 5): a ring of K search-direction buffers,
 * x += K terms every K-th iteration, (4 K + 1) / K update passes per iteration; 1 [default]: 2 on an unpartitioned lattice, 8 on a partitioned one (what was measured
 * faster in each case); 0: x every iteration),
 * halo_fold (1 [default], round 5: where the collective timing picks the one-stream halo schedule 3, the stencil launch takes the boundary hops from the ghost
 * buffers itself -- no exterior kernel, for every operator and both precisions; read-only halo_fold_active), cg_persist (1 [default]: a staggered CG on an unpartitioned lattice of
 * at most 256 chunks of 64 sites runs as ONE launch -- initial residual and all iterations, two grid-wide synchronisations per iteration, every wait bounded: if the workgroups are not all resident (a busy GPU) x is left untouched, the solve is
 * repeated by the launch chain and the key drops to 0; 0: the launch chain; 2: test hook that forces that fall-back), cg_small (1 [default]: on an unpartitioned lattice with <= 1024 stencil workgroups the two reduction launches of a fused CG
 * iteration are folded into the prologues of the kernels that consume them -- 3 dependent launches instead of 5, identical iterates), clover_fused (1 [default]: A x in the epilogue of the split kernel), clover_transport (1: partitioned-lattice
 * construction of the clover term / force also on one rank);
 * partitioned lattices: halo_merge (1 [default]: one message per peer when both faces go to the same rank), halo_stream_mode
 * (-1 [default]: time the four schedules of the halo exchange once -- collectively: every rank adopts the schedule with the smallest time summed over
 * the ranks; 0 | 1 | 2 force one of the overlapping schedules (exchange / interior / pack + exchange on the second stream), 3 forces pack -> exchange ->
 * interior -> exterior in order on one stream: no overlap and no cross-queue join, the faster one when the exchange is short), halo_tuned_us0..3 (read-only: the
 * times that choice was made from), halo_fuse (bit 1 [default 2]: the
 * exterior of D p packs the faces D^+ needs and the CG update packs the new search direction -- no separate pack launches; bit 0: the exterior's last
 * block sums the |.|^2 partials -- measured slower, off), staple_recon (1 [default]: the staple sweep reads two rows of links known to be on the group);
 * round 4: gauge_delta (1 [default]: a field that fails the 12-real gate but lies within 1e-9 of the group -- the reference's text / ILDG configurations -- is read by the
 * scalar-addressing Wilson kernel as rows 0, 1 in fp64 + the fp32 deviation of row 2, 128 B per link; results equal the 18-real kernel's to fp64 rounding; recon_active reads 2),
 * dslash_s18 (1 [default]: the scalar-addressing kernel also on the 18 stored reals), bicg_fused (even-odd BiCGStab of the Wilson / Wilson-clover operator: 2 inner
 * products from the Schur operator's epilogue and, up to 1024 chunks per parity, reductions and scalar steps in the consumers' prologues; 1 the same with separate reduction
 * launches -- bit-identical to 2; 4 [default] = 2 with the x / r and p updates merged into one launch on recurrences for rho' and |r'|^2 (equal to 2 up to rounding; bicg_rec_guard: digits of cancellation the |r'|^2 recurrence may show before the
 * stopping test waits for the summed value); 3 = 2 with a grid barrier between the two updates (slower); 0 the generic chain; read-only bicg_xrp_active: 0 | 1 (form 3) | 2 (form 4)), action_eo_solver (1 [default]: lqcd_fermi_action / lqcd_calc_UdSfdU / lqcd_action_* solve the Wilson(-clover) normal
 * equations as two even-odd BiCGStab solves under the reference's stopping rule; 0: CG), bicg_mixed (1: lqcd_solve_bicgstab_eo on the plain Wilson operator runs an fp32 inner
 * chain inside an fp64 defect correction, the stopping rule holds for the true fp64 residual; mixed_action_solver = 1 switches it on for the action solves), lazy_links (1: the per-direction link-call triples are recorded and fused -- the temporaries of a completed triple are then never written, so the C ABI's default is 0 (eager) and the Julia / Python bindings switch it on when they create a context,
 * see lqcd_link_*; read-only lazy_open, lazy_deferred), lazy_merge (1 [default]: a complete link update U <- exp(a P) U waits unlaunched and a second one of the same
 * fields, with nothing in between that reads U or writes P, adds its step -- the back-to-back half steps of runMD_QPQ_sw!, standardMD.jl:146-166).
 * Threads: calls on one context must not overlap (one lock per context in a host that uses several threads).  The exception is lqcd_gauge_destroy / lqcd_spinor_destroy,
 * which garbage collectors call from finalizer threads: with lazy_links on, a gauge-shaped field destroyed from another thread than the context's own (the creating thread,
 * or the one that last set adopt_thread = 1) is parked and freed by the context's thread at its next lqcd_gauge_create / lqcd_ctx_sync / lqcd_ctx_destroy (read-only
 * parked_fields counts them); slice views are counted under a lock. */
int lqcd_ctx_set_param(lqcd_ctx_t ctx, const char* key, int value);
int lqcd_ctx_get_param(lqcd_ctx_t ctx, const char* key, int* value);

/* RCCL bootstrap for one-process-per-GPU runs: rank 0 creates the id blob (two ncclUniqueIds: one communicator for halos,
 * one for reductions), the host broadcasts the 256 bytes (torch.distributed / MPI.jl), every rank calls
 * lqcd_ctx_comm_init.  Replaces the MPI.Init / PEs plumbing of src/mpi/mpimodule.jl:4-13. */
int lqcd_comm_unique_id(unsigned char id[256]);
int lqcd_ctx_comm_init(lqcd_ctx_t ctx, const unsigned char id[256], int nranks);
/* The second communication backend: peer-mapped windows (csrc/comm.hip; SURVEY.md 8(e) "or peer-mapped writes").  Every rank allocates one device
 * "window" (ghost buffers, mailboxes, flag words, reduction slots), lqcd_ctx_peer_export writes its 256-byte description (hipIpcMemHandle, process,
 * device, layout check), the host gathers the descriptions of all ranks in rank order (MPI.Allgather / torch.distributed.all_gather) and every rank
 * calls lqcd_ctx_peer_init with the nranks x 256 bytes.  From then on the pack kernels store faces straight into the neighbours' ghost buffers, the
 * exchange is a one-wave flag kernel and the solver's scalar sums travel through slots -- no RCCL launch anywhere.  One node (<= 8 ranks, same host);
 * ranks MAY share a device (the world-size-2 tests run two processes on one GPU).  Either this pair or lqcd_ctx_comm_init, not both.  Replaces the same
 * MPI plumbing of the reference (src/mpirun.jl:17-19, src/mpi/mpimodule.jl:4-13). */
#define LQCD_PEER_BLOB_BYTES 256
int lqcd_ctx_peer_export(lqcd_ctx_t ctx, unsigned char blob[LQCD_PEER_BLOB_BYTES]);
int lqcd_ctx_peer_init(lqcd_ctx_t ctx, const unsigned char* blobs, int nranks);
enum { LQCD_COMM_NONE = 0, LQCD_COMM_RCCL = 1, LQCD_COMM_PEER = 2 };
int lqcd_ctx_comm_backend(lqcd_ctx_t ctx, int* backend);
/* in-process emulation of a PE grid on ONE device (testing the halo path without RCCL): link `n` contexts that
 * were created with ranks 0..n-1 of the same pe_grid; afterwards use the lqcd_mdom_* collectives below. */
int lqcd_ctx_link_local(lqcd_ctx_t* ctxs, int n);

/* ---------------------------------------------------------------- gauge field  (Gaugefields.jl: Initialize_Gaugefields, universe.jl:41-49) */
int lqcd_gauge_create(lqcd_ctx_t ctx, lqcd_gauge_t* g);
int lqcd_gauge_destroy(lqcd_gauge_t g);
int lqcd_gauge_upload(lqcd_gauge_t g, const double* host, int layout);   /* substitute_U! / load_* (universe.jl:58-77) */
int lqcd_gauge_download(lqcd_gauge_t g, double* host, int layout);
/* the same for host arrays that carry the reference's wing of width nwing (Initialize_Gaugefields(NC, Nwing, ...), universe.jl:41-49;
 * test/test_wilson.toml has Nwing = 1): extents L + 2 nwing, only the interior is read / written */
int lqcd_gauge_upload_wing(lqcd_gauge_t g, const double* host, int nwing);
int lqcd_gauge_download_wing(lqcd_gauge_t g, double* host, int nwing);
int lqcd_gauge_unit(lqcd_gauge_t g);                                     /* condition = "cold" (universe.jl:41-49) */
int lqcd_gauge_hot_start(lqcd_gauge_t g, uint64_t seed);                  /* condition = "hot"; counter-based, keyed by GLOBAL site */
int lqcd_gauge_plaquette(lqcd_gauge_t g, double* plaq);                   /* calculate_Plaquette (lqcd.jl:187-193), normalised 1/(6 V NC) */
/* diagnostic, no reference counterpart: max over all links and elements of |row2 - conj(row0 x row1)| -- the quantity the 12-real link
 * path (tunable gauge_recon) is gated on (<= 1e-14 on every link); this rank's sub-lattice only */
int lqcd_gauge_unitarity_deviation(lqcd_gauge_t g, double* maxdev);

/* ---------------------------------------------------------------- fermion fields (Initialize_pseudofermion_fields, universe.jl:107,112) */
int lqcd_spinor_create(lqcd_ctx_t ctx, lqcd_spinor_t* s, int kind, int subset);
int lqcd_spinor_destroy(lqcd_spinor_t s);
int lqcd_spinor_upload(lqcd_spinor_t s, const double* host);    /* reference layout; FULL-lattice host array even for EVEN/ODD subsets */
int lqcd_spinor_download(lqcd_spinor_t s, double* host);        /* EVEN/ODD subsets write only their sites */
int lqcd_spinor_upload_wing(lqcd_spinor_t s, const double* host, int nwing);   /* fields created without nowing = true (universe.jl:107) */
int lqcd_spinor_download_wing(lqcd_spinor_t s, double* host, int nwing);
int lqcd_spinor_zero(lqcd_spinor_t s);                         /* clear_fermion! */
int lqcd_spinor_copy(lqcd_spinor_t dst, lqcd_spinor_t src);     /* substitute_fermion! */
int lqcd_spinor_gaussian(lqcd_spinor_t s, uint64_t seed);       /* gauss_distribution_fermion!: re,im ~ N(0,1), keyed by GLOBAL site */
int lqcd_spinor_z4(lqcd_spinor_t s, uint64_t seed);             /* Z4_distribution_fermi! (unusedfiles/measure_chiral_condensate.jl:180) */
int lqcd_spinor_point_source(lqcd_spinor_t s, const int global_xyzt[4], int ic, int is); /* setindex_global! (measure_Pion_correlator.jl:376) */
/* even/odd halves of a FULL spinor <-> EVEN/ODD spinors */
int lqcd_spinor_extract(lqcd_spinor_t half, lqcd_spinor_t full);
int lqcd_spinor_insert(lqcd_spinor_t full, lqcd_spinor_t half);

/* BLAS-1 (LinearAlgebra.dot / add_fermion! on fermion fields; standardHMC.jl:54).
 * dot is the Hermitian inner product sum conj(a) b; results are global (all-reduced over ranks). */
int lqcd_dot(lqcd_spinor_t a, lqcd_spinor_t b, double* re, double* im);
int lqcd_norm2(lqcd_spinor_t a, double* n2);
int lqcd_axpy(double ar, double ai, lqcd_spinor_t x, lqcd_spinor_t y);                         /* y += a x */
int lqcd_axpby(double ar, double ai, lqcd_spinor_t x, double br, double bi, lqcd_spinor_t y); /* y = a x + b y  (add_fermion!) */
int lqcd_scale(double ar, double ai, lqcd_spinor_t x);

/* ---------------------------------------------------------------- Dirac operator (Dirac_operator(U,x,params), universe.jl:137) */
/* kappa_or_mass: "kappa" (Wilson) or "mass" (staggered); r: Wilson parameter; bc: "boundarycondition" (+-1 per direction,
 * default [1,1,1,-1], parameter_structs.jl:133) */
int lqcd_op_create(lqcd_ctx_t ctx, lqcd_op_t* op, int kind, lqcd_gauge_t g, double kappa_or_mass, double r,
                   const int bc[4]);
int lqcd_op_destroy(lqcd_op_t op);
/* Dirac_operator = "WilsonClover", Clover_coefficient (src/system/parameter_structs.jl:125, test/test_wilsonclover.toml:9; the
 * reference rejects the operator, universe.jl:129-131, so this is the textbook definition): D_sw = D + i kappa c_sw sum_{mu<nu}
 * sigma_{mu nu} F_{mu nu}.  The term follows the links of the operator's gauge field; csw = 0 switches it off.  Supported by
 * lqcd_op_apply / _DdagD, the CG, BiCGStab, even-odd BiCGStab (inverse clover blocks), multi-shift and mixed-precision solvers,
 * also on a partitioned lattice (RCCL ranks; the clover sums are built with two matrix-face exchanges), and by the fermion
 * force (lqcd_fermion_force / lqcd_calc_UdSfdU add the derivative of the clover term; on a partitioned lattice through a
 * halo-extended copy of the links and Lambda matrices, RCCL ranks only). */
int lqcd_op_set_clover(lqcd_op_t op, double csw);
int lqcd_op_set_gauge(lqcd_op_t op, lqcd_gauge_t g);   /* the D(U) rebind idiom (unusedfiles/measure_chiral_condensate.jl:173) */
/* Dirac_operator = "Domainwall" (src/system/universe.jl:116-128: params "mass" = Domainwall_m, "L5", "M" = Domainwall_M; test/test_domainwallhmc.toml;
 * the fifth HMC fermion test of test/runtests.jl:132-137).  The arithmetic lives in LatticeDiracOperators.jl (not under the reference tree): this is the
 * textbook Shamir operator in the library's Wilson conventions [EXT-RECALL, parity unpinned -- csrc/domainwall.hip states it],
 *     (D5 psi)(s) = D4 psi(s) + psi(s) - P_- psi(s+1) - P_+ psi(s-1),  psi(L5+1) := -m psi(1), psi(0) := -m psi(L5),  D4 = Wilson operator of mass M (r = 1).
 * Fields are five-dimensional (lqcd_spinor_create_5d; lqcd_spinor_slice hands out the reference's x.w[i5] as Wilson fields that alias the slices: upload,
 * download and fills go through them, BLAS-1 takes the whole field).  Served for this operator: lqcd_op_apply, lqcd_op_apply_DdagD, lqcd_solve_cg_DdagD
 * and the pseudofermion action -- S = phi^+ D_PV (D^+D)^-1 D_PV^+ phi with the Pauli-Villars operator D_PV = D5(m = 1) -- through lqcd_action_* (heat
 * bath, action, force).  Tunable dw_batched (1 [default]: an application is one launch over all slices where the scalar-addressing Wilson kernel applies; read-only
 * dw_active).  RCCL ranks as the Wilson operator (no in-process PE grid); every other entry point answers LQCD_ERR_UNSUPPORTED. */
int lqcd_op_create_domainwall(lqcd_ctx_t ctx, lqcd_op_t* op, lqcd_gauge_t g, double M, double mass, int L5, const int bc[4]);
int lqcd_spinor_create_5d(lqcd_ctx_t ctx, lqcd_spinor_t* s, int L5);      /* Initialize_pseudofermion_fields(U[1], "Domainwall", L5 = L5) (universe.jl:128) */
int lqcd_spinor_slice(lqcd_spinor_t s5, int i5, lqcd_spinor_t* view);     /* 0-based; the view owns nothing; a parent destroyed first keeps its storage until its last view is destroyed */
/* mul!(y, D, x) / mul!(y, D', x) on FULL spinors */
int lqcd_op_apply(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger);
/* mul!(y, DdagD_operator(D), x):  out = D^dagger D in  (tmp is library scratch) */
int lqcd_op_apply_DdagD(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in);
/* parity hop: out (EVEN|ODD subset) = H in, in of the opposite subset; Wilson H = sum_nu[(r-g)U x+ + (r+g)U^+ x-],
 * staggered H = 1/2 sum_nu eta(U x+ - U^+ x-) */
int lqcd_op_hop(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger);

/* ---------------------------------------------------------------- solvers (solve_DinvX!) */
/* Stopping rule real(r.r) < eps (absolute, squared; default eps_CG = 1e-19, MaxCGstep = 3000,
 * parameter_structs.jl:174-175).  x holds the initial guess on entry.  iters/final_rr may be NULL. */
int lqcd_solve_cg_DdagD(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, double eps, int maxiter, int* iters,
                        double* final_rr);                       /* solve_DinvX!(y, DdagD, x) -> cg (AbstractMD.jl:129 via calc_UdSfdU!) */
/* staggered only: the parity block (D^+D)_pp = m^2 - H_pq H_qp of the block-diagonal D^+D, solved for the parity-p (0 even, 1 odd)
 * halves of the FULL fields x and b with half-lattice vectors; the other half of x is not touched.  The solve behind the
 * reference's 4-taste staggered action (pseudofermion on the even sites, test/test_staggered.toml); lqcd_fermi_action and
 * lqcd_calc_UdSfdU take it by themselves when the odd half of eta is exactly zero (tunable staggered_parity_solve). */
int lqcd_solve_cg_DdagD_parity(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int parity, double eps, int maxiter, int* iters,
                               double* final_rr);
int lqcd_solve_bicgstab(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter,
                        int* iters, double* final_rr);           /* solve_DinvX!(y, D | D', x) (standardHMC.jl:71) */
int lqcd_solve_bicg(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter, int* iters,
                    double* final_rr);                               /* method_CG = "bicg", the reference's default: host-scalar BiCG with D and D^+ */
int lqcd_solve_bicgstab_eo(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter,
                           int* iters, double* final_rr);        /* even-odd preconditioned variant (Wilson; with a clover term the
                                                                  * Schur complement uses the inverse clover blocks, built on first use) */
/* shiftedcg(vec_x, vec_beta, x, A, b) of the RHMC path (README.md:132; test/test_Nf2.toml:8, test_Nf3.toml:8):
 * (D^+D + sigma_j) xs[j] = b for all j < ns, plus the unshifted solution x0 (may be NULL), from ONE Krylov space.
 * Zero initial guesses; stops when the residual of the UNSHIFTED system obeys |r|^2 < eps (every shifted residual zeta_j^2 |r|^2 is
 * smaller; a shift whose own residual has reached eps is frozen earlier). */
int lqcd_solve_multishift_cg(lqcd_op_t op, lqcd_spinor_t x0, lqcd_spinor_t* xs, lqcd_spinor_t b, const double* sigma, int ns,
                             double eps, int maxiter, int* iters, double* final_rr);
/* mixed-precision CG on D^+D (SURVEY.md 8(b) "later solve_mixed_cg", 8(f) rank 3; BASELINE configs[4]): fp32 inner CG (fp32
 * copies of the links and work vectors, fp32 build of the stencil) inside an fp64 defect correction.  Same contract as
 * lqcd_solve_cg_DdagD -- solve_DinvX!(y, DdagD, x) -- except that the stopping rule real(r.r) < eps is enforced on the TRUE
 * residual b - D^+D x recomputed in fp64.  x holds the initial guess.  inner_tol: relative residual asked of each fp32 solve
 * (<= 0: chosen per step -- as few defect-correction steps as an fp32 recurrence supports, one per 1e-6 of the residual norm still to go, the
 * required reduction split evenly over them).  iters: total inner iterations (+ fp64 iterations if the fall-back ran); outer: correction steps. */
int lqcd_solve_mixed_cg_DdagD(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, double eps, int maxiter, double inner_tol, int* iters,
                              int* outer, double* final_rr);
/* diagnostic, no reference counterpart: out = D in (D^+ in) through the fp32 operator the mixed-precision solvers use on this operator (fp32
 * links, fp32 kernel and field layout; tunable mixed_pair32 selects the site-pair kernel where it applies), converted back to fp64;
 * reps > 0 also returns the mean time of one fp32 application in ms */
int lqcd_op_apply_f32(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int reps, double* ms);

/* mixed-precision shiftedcg (BASELINE configs[4] "RHMC ... mixed-precision fp32 inner / fp64 outer CG"; the
 * reference's shiftedcg, README.md:132, is fp64 throughout).  Same contract as lqcd_solve_multishift_cg -- zero initial guesses, x0
 * may be NULL -- with the stopping rule enforced on the TRUE fp64 residual of every system: |b - (D^+D + sigma_j) xs[j]|^2 < eps.
 * Phase 1: one fp32 multi-shift CG for all systems (relative residual inner_tol, <= 0: 1e-6); phase 2: fp64 defect correction of each
 * system on its own with fp32 one-shift solves (a multi-shift recurrence cannot be restarted).  iters: all fp32 iterations (+ fp64 ones
 * of fall-backs); outer: fp32 correction solves; final_rr: the largest true residual. */
int lqcd_solve_multishift_mixed_cg(lqcd_op_t op, lqcd_spinor_t x0, lqcd_spinor_t* xs, lqcd_spinor_t b, const double* sigma, int ns,
                                   double eps, int maxiter, double inner_tol, int* iters, int* outer, double* final_rr);

/* ---------------------------------------------------------------- pseudofermion action and force */
/* evaluate_FermiAction(fa, U, eta) (src/updates/standardHMC.jl:71): S_f = eta^+ (D^+D)^-1 eta by CG from a zero guess.
 * X receives (D^+D)^-1 eta; Y (may be NULL) receives D X.  Both stay on the device for lqcd_fermion_force. */
int lqcd_fermi_action(lqcd_op_t op, lqcd_spinor_t eta, lqcd_spinor_t X, lqcd_spinor_t Y, double eps, int maxiter, double* Sf,
                      int* iters);
/* the outer-product sweep of calc_UdSfdU!(UdSfdU, fa, U, eta) (src/md/AbstractMD.jl:129): out_mu(n) = "U dS_f/dU", a general
 * 3x3 matrix per link in a gauge-shaped field, DEFINED by
 *     d/d eps S_f[ U_mu(n) -> exp(i eps T) U_mu(n) ] = -2 Im tr( T out_mu(n) )      for every Hermitian T
 * (checked against finite differences of S_f in the tests).  Download with lqcd_gauge_download.  Collective on a partitioned
 * lattice: one exchange of the lower-face X, Y spinors (RCCL) feeds the links of the upper faces. */
int lqcd_fermion_force(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t X, lqcd_spinor_t Y);
/* the same sequence on an in-process PE grid (lqcd_ctx_link_local; tests): arrays ordered by rank */
int lqcd_mdom_fermion_force(int n, lqcd_op_t* ops, lqcd_gauge_t* outs, lqcd_spinor_t* X, lqcd_spinor_t* Y);
/* calc_UdSfdU! in one call: solve, Y = D X and the sweep, all resident; Sf and iters may be NULL */
int lqcd_calc_UdSfdU(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t eta, double eps, int maxiter, double* Sf, int* iters);
/* lqcd_fermion_force with out = (accumulate ? out : 0) + scale * G -- sums over the poles of a rational action in place */
int lqcd_fermion_force_acc(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t X, lqcd_spinor_t Y, double scale, int accumulate);
/* General-Nf (RHMC) pseudofermion action of the reference's staggered runs (FermiAction(D, Dict("Nf" => 2 | 3)),
 * src/system/universe.jl:109,138; test/test_Nf2.toml:8, test/test_Nf3.toml:8; README.md:132):
 * S_f = phi^+ (D^+D)^(-Nf/8) phi with the partial fractions  x^(-alpha) ~= a0 + sum_k res[k] / (x + poles[k])  supplied by the host
 * (poles >= 0).  One multi-shift solve per call, the shifted solutions stay in the context's scratch pool.
 *   lqcd_rational_apply:  y = a0 x + sum_k res[k] (D^+D + poles[k])^-1 x     (evaluate_FermiAction: S_f = Re <phi, y>;
 *                         heat bath of sample_pseudofermions!: phi = D^+D y with the fit of x^(Nf/16 - 1))
 *   lqcd_rational_force:  out = sum_k res[k] G[X_k, D X_k], X_k = (D^+D + poles[k])^-1 phi  -- calc_UdSfdU! for this action, same
 *                         convention as lqcd_fermion_force.  iters may be NULL. */
int lqcd_rational_apply(lqcd_op_t op, lqcd_spinor_t y, lqcd_spinor_t x, double a0, int n, const double* res, const double* poles,
                        double eps, int maxiter, int* iters);
int lqcd_rational_force(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t phi, int n, const double* res, const double* poles, double eps,
                        int maxiter, int* iters);

/* The coefficients themselves (host-only numerics, no GPU needed): x^(-alpha) ~= a0 + sum_k res[k] / (x + poles[k]) on [lam_min, lam_max],
 * 0 < alpha < 1, a0 >= 0, res[k] > 0, poles[k] > 0, max relative error <= tol on the interval (verified on a grid that is not the fit grid;
 * returned in max_rel_err, may be NULL).  res / poles must hold max_poles entries, *n receives the count.  The reference's package ships Remez
 * tables for its staggered Nf = 2 / 3 runs (test/test_Nf2.toml:8, test/test_Nf3.toml:8, README.md:112,132); here the fit is computed (AAA
 * algorithm, csrc/rational.hip).  LQCD_ERR_NOT_CONVERGED: the requested accuracy is not reachable in double precision on this interval. */
int lqcd_rational_fit(double alpha, double lam_min, double lam_max, double tol, int max_poles, double* a0, double* res, double* poles, int* n,
                      double* max_rel_err);
/* extreme Ritz values of D^+D from `steps` Lanczos iterations on the device: theta_max converges to the largest eigenvalue from below, theta_min to
 * the smallest from above -- use them with a margin.  The Wilson rational action takes its fit interval from here when none is given. */
int lqcd_estimate_spectrum(lqcd_op_t op, int steps, uint64_t seed, double* theta_min, double* theta_max);
/* The host half of the residual bound behind the rational actions' fit interval (lanczos_steps_used / ritz_bound of lqcd_action_get): the index-th
 * eigenvalue (ascending, 0-based) of the symmetric n x n tridiagonal (diag[n], offdiag[n-1]) and |last component| of its normalised eigenvector.
 * With the Lanczos coefficients of D^+D, beta_n * last_component bounds the distance from theta to an eigenvalue of D^+D.  Host-only, no GPU. */
int lqcd_tridiag_ritz(int n, const double* diag, const double* offdiag, int index, double* theta, double* last_component);

/* ---------------------------------------------------------------- FermiAction(D, Dict("Nf" => n)) as a handle (src/system/universe.jl:106-110,138)
 * nf <= 0: the operator's default (Wilson 2, staggered 4).  Wilson Nf = 2 and staggered Nf = 8 (eta on every site) / 4 (eta on the even sites,
 * test/test_staggered.toml) are the exact actions S_f = eta^+ (D^+D)^-1 eta; staggered 0 < Nf < 8 otherwise (test/test_Nf2.toml:8, test/test_Nf3.toml:8,
 * test/runtests.jl:114-130) and Wilson(-clover) 0 < Nf < 2 are the rational action S_f = eta^+ (D^+D)^(-Nf/n0) eta, n0 = 8 | 2, with partial fractions
 * fitted at creation on [m^2, m^2 + 16] (staggered) or on a Lanczos estimate of the spectrum with margins 0.5 / 1.2 (Wilson).  eps / maxiter:
 * "eps_CG" / "MaxCGstep" of the operator's parameters (universe.jl:103-135).  Optional parameters as nparams (key, value) pairs:
 * rhmc_lambda_min, rhmc_lambda_max (fix the interval), rhmc_tol_action (1e-12: action and heat bath), rhmc_tol_MD (1e-8: force),
 * rhmc_lanczos_steps (60), force_rational (1: staggered Nf = 4 through partial fractions, a statistical cross-check).
 * The operator must outlive the action.  U (may be NULL = the operator's links) is the `U` argument of the reference's generics: the operator is
 * rebound to it first (the D(U) idiom). */
int lqcd_action_create(lqcd_op_t op, double nf, double eps, int maxiter, int nparams, const char* const* keys, const double* values,
                       lqcd_action_t* fa);
int lqcd_action_destroy(lqcd_action_t fa);
int lqcd_action_set_solver(lqcd_action_t fa, double eps, int maxiter);   /* the operator's eps_CG / MaxCGstep changed after the action was made */
/* keys: rational, evensite, Nf, alpha, lambda_min, lambda_max, interval_refits, explicit_interval */
int lqcd_action_get(lqcd_action_t fa, const char* key, double* value);
/* which = 0: x^(-alpha) to rhmc_tol_action (the action), 1: the same to rhmc_tol_MD (the force), 2: x^(alpha/2 - 1) (heat bath).  res / poles may be
 * NULL to query the count */
int lqcd_action_coefficients(lqcd_action_t fa, int which, double* a0, double* res, double* poles, int capacity, int* n, double* max_rel_err);
int lqcd_action_set_coefficients(lqcd_action_t fa, int which, double a0, int n, const double* res, const double* poles);   /* e.g. Remez tables */
/* Wilson rational action: Lanczos on the current links; refits an estimated interval that has become too tight (interval_refits counts), raises
 * LQCD_ERR_ARG when a Ritz value left an interval the caller fixed.  Called by sample / evaluate themselves. */
int lqcd_action_check_interval(lqcd_action_t fa);
int lqcd_action_gauss_sampling(lqcd_action_t fa, lqcd_spinor_t xi, uint64_t seed);                           /* gauss_sampling_in_action!(xi, U, fa) (src/md/standardMD.jl:95): <|xi_i|^2> = 1 */
int lqcd_action_sample_pseudofermions(lqcd_action_t fa, lqcd_gauge_t U, lqcd_spinor_t eta, lqcd_spinor_t xi); /* sample_pseudofermions!(eta, U, fa, xi) (standardMD.jl:96) */
/* evaluate_FermiAction(fa, U, eta) (src/updates/standardHMC.jl:69,71).  X, Y may be NULL (library scratch); otherwise X receives (D^+D)^-1 eta and Y = D X
 * (exact actions) or X the rational image of eta */
int lqcd_action_evaluate(lqcd_action_t fa, lqcd_gauge_t U, lqcd_spinor_t eta, lqcd_spinor_t X, lqcd_spinor_t Y, double* Sf, int* iters);
/* calc_UdSfdU!(UdSfdU, fa, U, eta) (src/md/AbstractMD.jl:129): out = G in the convention of lqcd_fermion_force; Sf (exact actions only) and iters may be NULL */
int lqcd_action_force(lqcd_action_t fa, lqcd_gauge_t U, lqcd_gauge_t out, lqcd_spinor_t eta, double* Sf, int* iters);

/* ---------------------------------------------------------------- gauge side of the MD step
 * Momenta are traceless anti-Hermitian 3x3 matrices held in a gauge-shaped field (lqcd_gauge_create).  Conventions (fixed by
 * dH/dtau = 0, tested): K = -sum tr P^2 (= p.p/2 for P = i p_a T_a); dU/dtau = P U; S_g = -(beta/3) sum_plaq Re tr U_p; every
 * force field G ("U dS/dU") obeys dS/d eps[U -> exp(i eps T) U] = -2 Im tr(T G), so dP/dtau = TA(G). */
int lqcd_gauge_copy(lqcd_gauge_t dst, lqcd_gauge_t src);                  /* substitute_U!(Uold, U) (src/updates/standardHMC.jl:45) */
int lqcd_gauge_action(lqcd_gauge_t U, double beta, double* Sg);          /* -evaluate_GaugeAction/NC (standardHMC.jl:50) */
int lqcd_gauge_force(lqcd_gauge_t out, lqcd_gauge_t U, double beta);     /* calc_dSdUmu! + mul!(temp, U, dSdUmu) (src/md/AbstractMD.jl:108-109): -(beta/6) U * staples.  Collective on a
                                                                          * partitioned lattice: forward ghost links, then the lower staples of the upper faces (two RCCL steps, no corners) */
int lqcd_mdom_gauge_force(int n, lqcd_gauge_t* outs, lqcd_gauge_t* Us, double beta, double factor, int fuse); /* the same on an in-process PE grid (tests) */
/* Stout smearing of the links the fermion action sees, and the chain rule back to the thin links (src/system/universe.jl:147-171: CovNeuralnet(U),
 * STOUT_Layer(p.stout_loops, p.stout_ρ, U); src/md/standardMD.jl:192-227: calc_smearedU, calc_UdSfdU! on the smeared links, back_prop;
 * src/updates/standardHMC.jl:67-68).  One STOUT layer with the plaquette loop, Morningstar-Peardon's definition [EXT-RECALL: Gaugefields.jl is not under the
 * reference tree]:  U'_mu(n) = exp(-rho TA(U_mu(n) A_mu(n))) U_mu(n), A = the six staples (those of lqcd_gauge_force).  Several layers = several calls,
 * back-propagated in reverse order with the links each layer started from.  Force fields in the convention of lqcd_fermion_force.  Collective on RCCL
 * ranks (staple faces; the back-propagation reads a halo-extended block of links and N matrices). */
/* calculate_Polyakov_loop(U, temp1, temp2) (the Polyakov_loop measurement of every toml under test/; src/system/lqcd.jl:141 -> QCDMeasurements):
 * 1/(NC NX NY NZ) sum_x tr prod_t U_4(x, t), summed over ranks; the time direction must not be partitioned */
int lqcd_gauge_polyakov(lqcd_gauge_t U, double* re, double* im);
int lqcd_link_mul_adj(lqcd_gauge_t C, int mu_c, lqcd_gauge_t A, int mu_a, lqcd_gauge_t B, int mu_b);   /* mul!(C, A', B): C = A^+ B site by site (standardMD.jl:211) */
int lqcd_stout_smear(lqcd_gauge_t out, lqcd_gauge_t U, double rho);                     /* out != U */
int lqcd_stout_backprop(lqcd_gauge_t G, lqcd_gauge_t Gs, lqcd_gauge_t U, double rho);   /* G at the thin links U from Gs at the smeared links; G = Gs allowed */
int lqcd_momentum_add_ta(lqcd_gauge_t P, double factor, lqcd_gauge_t G); /* Traceless_antihermitian_add!(p, factor, G) (AbstractMD.jl:110,131) */
int lqcd_momentum_add_gauge_force(lqcd_gauge_t P, double factor, lqcd_gauge_t U, double beta); /* P_update! (AbstractMD.jl:99-118) fused: P += factor TA(gauge force), the force field is never stored.  P must hold traceless anti-Hermitian matrices (the reference's p[mu] is a TA field by type; every writer of the library -- lqcd_momentum_gaussian, the *_add_ta entries -- stores them exactly so): the sweep reads the upper triangle of P only */
int lqcd_gauge_exp_update(lqcd_gauge_t U, double dt, lqcd_gauge_t P);    /* U_update! (AbstractMD.jl:78-97): U <- exp(dt P) U (tunable md_reunitarize: projected back onto SU(3) in the same pass) */
int lqcd_gauge_reunitarize(lqcd_gauge_t U);                              /* no reference counterpart: every link back onto SU(3) (Gram-Schmidt rows 0,1; row 2 = conj(row0 x row1)); once per trajectory keeps the 12-real Dslash alive under per-direction callers */
int lqcd_momentum_gaussian(lqcd_gauge_t P, uint64_t seed);               /* gauss_distribution!(p) (src/md/standardMD.jl:86); keyed by GLOBAL site */
int lqcd_momentum_action(lqcd_gauge_t P, double* K);                     /* p.p/2 (standardHMC.jl:49) */

/* Single-direction forms: the reference's unchanged callers hold its Vector of link fields one direction at a time -- U[mu], p[mu]
 * and temporary link fields (src/md/AbstractMD.jl:78-135).  A "link field" here is (gauge-shaped field, direction slot 0..3); nothing
 * is copied to form one.  The four-direction calls above are the fused fast path of the same arithmetic. */
int lqcd_link_copy(lqcd_gauge_t dst, int mu_dst, lqcd_gauge_t src, int mu_src);          /* substitute_U!(U[mu], W) (AbstractMD.jl:93) */
int lqcd_link_scaled_copy(lqcd_gauge_t dst, int mu_dst, double s, lqcd_gauge_t src, int mu_src);  /* dst = s * src; s = -1 turns the force field G of lqcd_calc_UdSfdU into the
                                                                                           * "U dS_f/dU" = -G the caller's factor = -eps dtau expects (AbstractMD.jl:127-132) */
int lqcd_link_exp(lqcd_gauge_t E, int mu_e, double t, lqcd_gauge_t P, int mu_p);          /* exptU!(expU, t, p[mu], temps) (AbstractMD.jl:91): E = exp(t P) */
int lqcd_link_mul(lqcd_gauge_t C, int mu_c, lqcd_gauge_t A, int mu_a, lqcd_gauge_t B, int mu_b);   /* mul!(W, expU, U[mu]), mul!(temp1, U[mu], dSdUmu) (AbstractMD.jl:92,109) */
int lqcd_link_add_ta(lqcd_gauge_t P, int mu_p, double factor, lqcd_gauge_t G, int mu_g);  /* Traceless_antihermitian_add!(p[mu], factor, temp1) (AbstractMD.jl:110,131) */
int lqcd_link_staple(lqcd_gauge_t out, int mu_out, lqcd_gauge_t U, int mu, double beta);  /* calc_dSdUmu!(dSdUmu, gauge_action, mu, U) (AbstractMD.jl:108) for the plaquette
                                                                                           * action of universe.jl:92-95: (beta/2) * sum of the six staples.  Collective on a partitioned lattice */
/* The per-direction call triples of the reference's U_update! / P_update! (AbstractMD.jl:91-93, 108-110) as ONE pass each; the bindings reach
 * them by evaluating exptU! -> mul! -> substitute_U! and calc_dSdUmu! -> mul! -> Traceless_antihermitian_add! lazily, the callers stay unchanged:
 *   W[mu_w] = exp(t P[mu_p]) U[mu_u]                      (W = U, mu_w = mu_u: the in-place update of one direction, projected back onto SU(3)
 *                                                          under the rule of lqcd_gauge_exp_update when the tunable md_reunitarize is set)
 *   P[mu_p] += factor * TA(U[mu] * (beta/2) * sum of the six staples of direction mu) */
int lqcd_link_exp_mul(lqcd_gauge_t W, int mu_w, double t, lqcd_gauge_t P, int mu_p, lqcd_gauge_t U, int mu_u);
int lqcd_link_add_ta_staple(lqcd_gauge_t P, int mu_p, double factor, lqcd_gauge_t U, int mu, double beta);

/* the SURVEY.md 8(d) protocol: every application between its own HIP events, median and mean over reps */
int lqcd_bench_dslash_median(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int warm, int reps, double* median_ms,
                             double* mean_ms);
/* multi-GPU diagnostics (bench.py, N > 1): phase breakdown of one partitioned operator application -- ms[6] = pack, interior,
 * exchange (pack end -> halos received), idle wait for the exchange after the interior, exterior, total -- and the latency of
 * the one-double all-reduce (microseconds) */
int lqcd_bench_halo_phases(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int reps, double* ms);
int lqcd_bench_allreduce(lqcd_ctx_t ctx, int reps, double* us);

/* benchmarking window: exactly niter CG iterations, exit test disabled (SURVEY.md 8(d) timing protocol) */
int lqcd_solve_cg_DdagD_fixed(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int niter);

/* ---------------------------------------------------------------- timing (hipEvents on the library's compute stream) */
/* runs `warm` untimed + `reps` timed applications of D (or D^dagger) and returns the mean ms per application */
int lqcd_bench_dslash(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int warm, int reps,
                      double* ms_per_apply);
/* ms per CG iteration over a fixed window of niter iterations (after `warm` untimed iterations) */
int lqcd_bench_cg(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int warm, int niter, double* ms_per_iter);

/* CG session for externally timed windows (bench.py brackets these with its own barriers/clock):
 * begin = r = b - D^+D x, p = r; iterate = enqueue n iterations (exit test disabled) and wait; end = release scratch */
int lqcd_cg_session_begin(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b);
int lqcd_cg_session_iterate(lqcd_op_t op, int n);
int lqcd_cg_session_end(lqcd_op_t op);

/* ---------------------------------------------------------------- in-process multi-domain collectives (testing only) */
int lqcd_mdom_op_apply(int n, lqcd_op_t* ops, lqcd_spinor_t* outs, lqcd_spinor_t* ins, int dagger);
int lqcd_mdom_dot(int n, lqcd_spinor_t* a, lqcd_spinor_t* b, double* re, double* im);
int lqcd_mdom_plaquette(int n, lqcd_gauge_t* g, double* plaq);
int lqcd_mdom_solve_cg_DdagD(int n, lqcd_op_t* ops, lqcd_spinor_t* x, lqcd_spinor_t* b, double eps, int maxiter,
                             int* iters, double* final_rr);

#ifdef __cplusplus
}
#endif
#endif
