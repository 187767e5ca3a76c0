#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X: CG iterations/s and Dslash GFLOP/s, 32^3x64 SU(3) Wilson fp64.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one CG iteration on D^+D (one D, one D^+ application + the BLAS-1/reduction work) of the
fixed global lattice; for N > 1 the 4-D lattice is domain-decomposed over a PE grid (strong scaling), halos travel
over RCCL inside liblqcd_hip.so, and the control plane (rendezvous of the RCCL id, barriers, max-over-ranks) uses
torch.distributed with the gloo backend so that PyTorch never touches the GPU the library drives.
Inputs are synthetic and resident in HBM before the timed region: hot-start links (seed 111), Gaussian source (seed 112),
kappa = 0.141139, r = 1, BC = [1,1,1,-1].

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (Wilson Dslash kernel, algorithmic 960 B/site over HIP-event
time measured on the library's own stream) and, at N = 1, "cpu_baseline" (the oracle timed on the host, a bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
WILSON_FLOP_PER_SITE = 1320
WILSON_BYTES_PER_SITE = 960    # read psi 192 + 4 links 576 + write 192
KAPPA = 0.141139
CG_TRAFFIC = {}               # filled by measure_traffic(): PMC bytes per launch of the kernels of a CG iteration
KERNEL_NAMES = {0: "wilson_interior", 1: "wilson_dirsplit", 2: "wilson_hopsplit", 3: "wilson_hopsplit_persist", 4: "wilson_lanesplit",
                5: "wilson_dirsplit4", 6: "wilson_dirsplit_lds", 7: "wilson_pair4", 8: "wilson_dirsplit_both"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lattice", type=str, default="32,32,32,64")
    ap.add_argument("--dslash-reps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pe-grid", type=str, default="")
    ap.add_argument("--comm", choices=("auto", "peer", "rccl"), default="auto",
                    help="N > 1: communication backend -- peer = hipIpc-mapped windows (csrc/comm.hip), rccl = RCCL send/recv + all-reduce, "
                         "auto = peer, falling back to rccl (on every rank) if any rank cannot map its neighbours; reported in config.comm_backend")
    ap.add_argument("--set", action="append", default=[], help="library tunable key=value")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LQCD_BENCH_SHARE_DEVICE"):      # testing aid: every rank on device 0 -- the N > 1 path with REAL processes on a one-GPU box (peer-mapped backend only:
        local_rank = 0                                 # RCCL refuses two ranks on one device); the figures of such a run are not performance numbers
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    gL = tuple(int(v) for v in args.lattice.split(","))

    # load the HIP library first: the ROCm runtime it was built with (/opt/rocm) is the one this process uses
    import latticeqcd_jl_amd as lq
    lq.lib.lib()

    dist = None
    force_dist = bool(os.environ.get("LQCD_BENCH_FORCE_DIST"))   # testing aid: take the N > 1 control path at world size 1
    if world > 1 or force_dist:
        import torch  # noqa: F401  (control plane only)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    pe = tuple(int(v) for v in args.pe_grid.split(",")) if args.pe_grid else lq.pegrid.choose_pe_grid(gL, world)
    lat = lq.Lattice(gL, pe, rank, device=local_rank)
    for kv in args.set:
        k, v = kv.split("=")
        lat.set_param(k, int(v))
    comm_note = None
    if world > 1 or force_dist:
        import torch
        if args.comm in ("auto", "peer"):
            ok, err = 1, ""
            try:
                blobs = [None] * world
                dist.all_gather_object(blobs, lat.peer_export())      # the 256-byte window descriptions, in rank order
                lat.peer_init(blobs)
            except Exception as e:                                    # noqa: BLE001  (whatever it is, every rank must learn of it)
                ok, err = 0, str(e)
            t = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                if args.comm == "peer":
                    raise RuntimeError("--comm peer: a rank could not set up its peer-mapped windows" + (": " + err if err else ""))
                comm_note = "peer-mapped windows unavailable on some rank" + (" (rank %d: %s)" % (rank, err) if err else "") + ": fell back to RCCL"
                lat.close()                                           # a context that exported a window keeps to that backend: start over
                lat = lq.Lattice(gL, pe, rank, device=local_rank)
                for kv in args.set:
                    k, v = kv.split("=")
                    lat.set_param(k, int(v))
        if lat.comm_backend == "none":
            box = [lq.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            lat.comm_init(box[0])
        _flush_c_stdio()

    def device_sync():
        lat.sync()   # hipStreamSynchronize on the library's compute and communication streams

    U = lq.Gaugefields(lat)
    lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, __import__("ctypes").c_uint64(111)))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": (1, 1, 1, -1)})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    x, y = b.similar(), b.similar()
    V = gL[0] * gL[1] * gL[2] * gL[3]
    Vloc = V // world

    # ---- N > 1: before anything is timed, the halo path must reproduce the one-GPU value of |D b|^2 (same global problem).  Under --comm auto a peer-mapped backend
    # that fails it is replaced by RCCL on every rank -- the first run on real links must not report the speed of a wrong answer.
    selfcheck = None
    if world > 1 or force_dist:
        try:
            selfcheck = halo_selfcheck(lq, D, b, y, gL)
        except Exception as e:      # a wait of the peer-mapped backend that gave up (dead link, no peer access across devices): treated as a failed check, not as a crash
            selfcheck = {"norm2_Db": float("nan"), "expected": DB_NORM2_ONE_GPU.get(tuple(gL), float("nan")), "rel_diff": float("nan"), "ok": False, "error": str(e)[:300]}
        if selfcheck is not None and dist is not None and world > 1:      # one verdict for every rank: a fall-back must be taken by all of them or by none
            import torch
            flag = torch.tensor([1 if selfcheck["ok"] else 0], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                flag = flag.cuda()
            try:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    selfcheck["ok"] = False
            except Exception as e:
                selfcheck["agreement_error"] = str(e)[:200]
        if selfcheck is not None and not selfcheck["ok"] and lat.comm_backend == "peer" and args.comm == "auto":
            comm_note = "peer-mapped windows gave |D b|^2 = %.17e, expected %.17e%s: fell back to RCCL" % (
                selfcheck["norm2_Db"], selfcheck["expected"], (" (" + selfcheck["error"] + ")") if "error" in selfcheck else "")
            for o in (x, y, b, D, U):
                o.close()
            lat.close()
            lat = lq.Lattice(gL, pe, rank, device=local_rank)
            for kv in args.set:
                k, v = kv.split("=")
                lat.set_param(k, int(v))
            box = [lq.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            lat.comm_init(box[0])
            U = lq.Gaugefields(lat)
            lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, __import__("ctypes").c_uint64(111)))
            D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": (1, 1, 1, -1)})
            b = lq.Fermionfields(lat, lq.WILSON)
            lq.gauss_distribution_fermion_(b, 112)
            x, y = b.similar(), b.similar()
            selfcheck = halo_selfcheck(lq, D, b, y, gL)

    # ---- Dslash kernel timing (HIP events on the stream the kernel is launched on)
    def settle(op=None, n=600):
        # untimed: ~0.2-0.3 s of the kernel itself, so that a timing which follows an idle stretch (start-up, the --pmc child processes)
        # starts at steady clocks -- r04: the first figure after the PMC passes read 0.52 ms for a 0.41 ms kernel
        lq.bench_dslash(op or D, y, b, warm=n, reps=1)

    barrier()
    settle()
    ms_dslash = lq.bench_dslash(D, y, b, warm=20, reps=args.dslash_reps)              # mean over back-to-back launches
    ms_median, _ = lq.bench_dslash_median(D, y, b, warm=5, reps=args.dslash_reps)       # per-launch events, median
    barrier()
    if dist is not None:
        import torch
        t = torch.tensor([ms_dslash], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_dslash = float(t.item())
    dslash_gflops = WILSON_FLOP_PER_SITE * V / (ms_dslash * 1e-3) / 1e9
    achieved = WILSON_BYTES_PER_SITE * Vloc / (ms_dslash * 1e-3) / 1e9     # GB/s per GPU, algorithmic bytes (960 B/site)
    recon_active = lat.get_param("recon_active")                            # 1: the kernel read 12 of the 18 reals of every link
    moved_per_site = 768 if recon_active else WILSON_BYTES_PER_SITE

    # ---- CG window: W warm-up + exactly K timed iterations, exit test disabled
    sess = lq.CGSession(D, x, b)
    sess.iterate(args.warmup)
    device_sync(); barrier()
    t0 = time.perf_counter()
    sess.iterate(args.steps)
    device_sync(); barrier()
    dt = time.perf_counter() - t0
    sess.close()
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    iters_per_s = args.steps / dt

    # ---- roofline.traffic: HBM/fabric bytes of ONE launch of the dominant kernel, measured in THIS run: two separate rocprofv3
    # --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) over a child process that applies the same operator to the same
    # synthetic configuration, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request of a 16-B/lane
    # read on gfx950: doubled; both counters are KiB).  N = 1 only; null with a reason when rocprofv3 is unavailable.
    traffic, traffic18, traffic_source = None, None, "not measured (N > 1)"
    if world == 1 and not force_dist:
        traffic, traffic18, traffic_source = (None, None, "skipped (--no-pmc)") if args.no_pmc else measure_traffic(args)
        settle()

    out = {
        "metric": "CG iters/sec (D^+D) & Dslash GFLOP/s, %d^3x%d SU(3) Wilson fp64" % (gL[0], gL[3]),
        "value": iters_per_s,
        "unit": "iter/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (hot-start SU(3) links seed 111, Gaussian source seed 112, kappa=0.141139, BC=[1,1,1,-1])",
        "config": {"workload": "configs[3]: %dx%dx%dx%d Wilson D^+D CG (fixed-length window), fp64" % gL,
                   "pe_grid": list(pe), "local_lattice": list(lat.local_L), "dslash_variant": lat.get_param("dslash_variant"),
                   "xcd_remap": lat.get_param("xcd_remap"), "xcd_nsub": lat.get_param("xcd_nsub"),
                   "xcd_ysplit": lat.get_param("xcd_ysplit"), "cg_fused": lat.get_param("cg_fused"),
                   "gauge_recon": lat.get_param("gauge_recon"), "gauge_recon_active": recon_active,
                   "comm_backend": lat.comm_backend, "comm_requested": args.comm, "comm_note": comm_note,
                   "ranks_share_device_0": bool(os.environ.get("LQCD_BENCH_SHARE_DEVICE"))},
        "dslash_gflops": dslash_gflops,
        "dslash_ms": ms_dslash,
        "dslash_ms_median_per_launch_events": ms_median,
        "roofline": {"bound": "hbm", "kernel": kernel_name(lat, recon_active) + " (mul!(y,D,x))", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_over_bytes_moved": (traffic / (moved_per_site * Vloc)) if traffic else None,
                     "algorithmic_bytes_per_site": WILSON_BYTES_PER_SITE, "sites_per_launch": Vloc,
                     # the default kernel rebuilds the third row of every (unitary) link: it MOVES 768 B/site for the 960 algorithmic ones
                     "compulsory_bytes_moved_per_site": moved_per_site,
                     "frac_by_bytes_moved": moved_per_site * Vloc / (ms_dslash * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }

    # ---- the CG iteration's own roofline: what the fused iteration MOVES (the SURVEY 8(d) figure of 4224 B/site is the UNFUSED model -- 2 x 960 + 12 spinor
    # passes -- and would read as ~1.0 x peak against ms_per_step; the fused iteration writes no q = D^+D p, updates r in D^+'s epilogue, x every second iteration,
    # and reads 12 of the 18 reals of a link)
    link_b = 384 if recon_active else 576
    per_kernel_moved = {"D": 192 + link_b + 192, "Ddag_update_mode": 192 + link_b + 192 + 192, "cg_update_even": 3 * 192, "cg_update_odd": 6 * 192}
    moved_iter = (per_kernel_moved["D"] + per_kernel_moved["Ddag_update_mode"] + 0.5 * (per_kernel_moved["cg_update_even"] + per_kernel_moved["cg_update_odd"])) * Vloc
    t_iter = dt / args.steps
    cg_traffic = None
    if CG_TRAFFIC and all(CG_TRAFFIC.get(k) for k in per_kernel_moved):
        cg_traffic = CG_TRAFFIC["D"] + CG_TRAFFIC["Ddag_update_mode"] + 0.5 * (CG_TRAFFIC["cg_update_even"] + CG_TRAFFIC["cg_update_odd"])
    out["cg_roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                          "bytes_moved_per_site_per_kernel": per_kernel_moved, "bytes_moved_per_iteration": moved_iter,
                          "achieved": moved_iter / t_iter / 1e9, "frac": moved_iter / t_iter / 1e9 / HBM_PEAK_GBS,
                          "traffic_per_kernel": dict(CG_TRAFFIC) if CG_TRAFFIC else None, "traffic": cg_traffic,
                          "frac_by_traffic": (cg_traffic / t_iter / 1e9 / HBM_PEAK_GBS) if cg_traffic else None,
                          "survey_unfused_model_bytes_per_site": 4224,
                          "note": "frac = bytes the fused iteration must move / ms_per_step / peak; roofline.frac (Dslash) is on SURVEY 8(d)'s 960 algorithmic B/site "
                                  "while the default kernel moves 768 (roofline.frac_by_bytes_moved)"}
    out["roofline"]["note"] = "frac is on the 960-B/site algorithmic model of SURVEY 8(d); the default kernel rebuilds row 2 of every link and moves 768 B/site (frac_by_bytes_moved)"

    # ---- secondary, N > 1 (outside the timed region): where the time of a partitioned operator application goes on real links
    if (world > 1 or force_dist) and any(p > 1 for p in pe) or (force_dist and os.environ.get("LQCD_FORCE_PARTITION")):
        import ctypes as C
        import torch
        ph = (C.c_double * 6)()
        us = C.c_double(0)
        barrier()
        st1 = lq.lib.lib().lqcd_bench_halo_phases(D._h, y._h, b._h, 0, 20, ph)     # diagnostics must never cost the bench line:
        barrier()                                                                  # a failing rank reports NaN and still joins
        st2 = lq.lib.lib().lqcd_bench_allreduce(lat._h, 200, C.byref(us))          # the reductions below
        vals = (list(ph) if st1 == 0 else [float("nan")] * 6) + [us.value if st2 == 0 else float("nan")]
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        v = [None if float(z) != float(z) else float(z) for z in t]   # NaN (a failed diagnostic) -> null, keeps the line valid JSON
        folded = bool(lat.get_param("halo_stream_mode") == 3 and lat.get_param("halo_fold") and v[4] == 0.0)
        out["halo_phases_ms_max_over_ranks"] = {"pack": v[0], "interior": v[1], "exchange_after_pack": v[2],
                                                "idle_wait_for_exchange": v[3], "exterior": v[4], "total_synchronised": v[5],
                                                # inside the fused CG the pack launch of a folded application is gone too: the x/p update and the reduction launch write the faces
                                                "schedule": "folded: pack -> exchange -> one stencil launch reading the ghost buffers (no exterior kernel)" if folded
                                                else "pack -> exchange || interior -> exterior",
                                                "pack_launches_per_cg_iteration": 0 if lat.get_param("halo_fuse") & 2 else 2,
                                                "exterior_launches_per_cg_iteration": 0 if folded else 2}
        out["allreduce_latency_us"] = v[6]
        out["halo_selfcheck"] = selfcheck      # |D b|^2 of the partitioned operator against the one-GPU value of the same global problem
        face = [lat.local_L[0] * lat.local_L[1] * lat.local_L[2] * lat.local_L[3] // lat.local_L[mu] if pe[mu] > 1 else 0 for mu in range(4)]
        out["halo_bytes_per_peer_and_direction"] = [96 * f for f in face]
        try:        # which schedule the library's one-off timing picked on rank 0 (0: exchange on the 2nd stream, 1: interior on it, 2: pack + exchange on it with the interior enqueued first, 3: one stream, no overlap, no join, 4: one stream, bulk in front of the exchange step and boundary behind it)
            out["halo_stream_mode_rank0"] = {"chosen": lat.get_param("halo_stream_mode"),
                                             "us_per_application": [lat.get_param("halo_tuned_us%d" % m) for m in range(5)]}
        except Exception:
            pass

    # ---- secondary (outside the timed region, not part of `value`): the same operator with all 18 stored reals of every link read
    # (gauge_recon = 18; what the default falls back to when a field is not unitary to 1e-14): 960 B/site moved.
    if world == 1 and not force_dist:
      recon0 = lat.get_param("gauge_recon")
      try:
        lat.set_param("gauge_recon", 18)
        settle()
        ms18 = lq.bench_dslash(D, y, b, warm=20, reps=args.dslash_reps)
        msi18 = lq.bench_cg(D, x, b, warm=5, niter=50)
        out["gauge_recon18_all_reals_read"] = {"recon_active": lat.get_param("recon_active"), "dslash_ms": ms18,
                                               "dslash_gflops": WILSON_FLOP_PER_SITE * V / (ms18 * 1e-3) / 1e9,
                                               "moved_bytes_per_site": WILSON_BYTES_PER_SITE,
                                               "frac_of_peak": WILSON_BYTES_PER_SITE * Vloc / (ms18 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "traffic": traffic18,
                                               "traffic_over_bytes_moved": (traffic18 / (WILSON_BYTES_PER_SITE * Vloc)) if traffic18 else None,
                                               "cg_iters_per_s": 1e3 / msi18}
      except Exception as e:                               # secondary numbers never cost the bench line
        out["gauge_recon18_all_reals_read"] = {"error": str(e)}
      finally:
        lat.set_param("gauge_recon", recon0)

    # ---- secondary (outside the timed region): links as the reference's own files hold them -- 11 significant digits, unitary to ~1e-10, so the 12-real gate
    # (1e-14) fails.  The scalar-addressing kernel then reads rows 0, 1 in fp64 + the fp32 deviation of row 2 ("12 + delta", 896 B/site moved, recon_active 2).
    if world == 1 and not force_dist:
      try:
        import numpy as _np
        Uh = U.download()
        rng = _np.random.default_rng(5)
        Uh += 1e-10 * (rng.standard_normal(Uh.shape) + 1j * rng.standard_normal(Uh.shape)) / 3.0
        U3 = lq.Gaugefields(lat).upload(Uh)
        del Uh
        D3 = lq.Dirac_operator(U3, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": (1, 1, 1, -1)})
        ms3 = lq.bench_dslash(D3, y, b, warm=20, reps=args.dslash_reps)
        act3 = lat.get_param("recon_active")
        msi3 = lq.bench_cg(D3, x, b, warm=5, niter=50)
        lat.set_param("gauge_delta", 0)
        ms3b = lq.bench_dslash(D3, y, b, warm=20, reps=args.dslash_reps)
        lat.set_param("gauge_delta", 1)
        out["reference_format_links"] = {"max_unitarity_deviation": lq.unitarity_deviation(U3), "recon_active": act3, "dslash_ms": ms3,
                                         "frac_of_peak_960B": WILSON_BYTES_PER_SITE * Vloc / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, "moved_bytes_per_site": 896 if act3 == 2 else 960,
                                         "cg_iters_per_s": 1e3 / msi3, "dslash_ms_all_18_reals": ms3b}
        D3.close(); U3.close()
      except Exception as e:
        out["reference_format_links"] = {"error": str(e)}

    # ---- secondary (outside the timed region, not part of `value`): time to solution r.r < 1e-16, fp64 CG vs mixed-precision CG
    if world == 1 and not force_dist:
      try:
        D.eps_CG = 1e-16
        A = lq.DdagD_operator(D)
        tts = {}
        for name, fn in (("fp64", lambda: lq.solve_DinvX_(x, A, b, return_info=True)),
                         ("mixed", lambda: lq.solve_mixed_DinvX_(x, A, b, return_info=True))):
            lq.clear_fermion_(x); fn()                      # warm (work-space allocation)
            lq.clear_fermion_(x); device_sync()
            t0 = time.perf_counter(); info = fn(); device_sync()
            tts[name] = (1e3 * (time.perf_counter() - t0), info)
        out["time_to_solution_1e-16"] = {"fp64_cg_ms": tts["fp64"][0], "fp64_iters": tts["fp64"][1][0],
                                         "mixed_cg_ms": tts["mixed"][0], "mixed_inner_iters": tts["mixed"][1][0],
                                         "mixed_outer_steps": tts["mixed"][1][1], "mixed_true_rr": tts["mixed"][1][2],
                                         "speedup": tts["fp64"][0] / tts["mixed"][0]}
      except Exception as e:
        out["time_to_solution_1e-16"] = {"error": str(e)}

    # ---- secondary (outside the timed region): the same CG window on links that went through molecular dynamics -- a copy of the field, 20 MD
    # steps of the reference's Sexton-Weingarten gauge legs (standardMD.jl:146-166, N = 10, dtau = 0.05: 400 link updates from a Gaussian
    # momentum), which is where the 12-real criterion (unitary to 1e-14) would be lost without the projection inside the link update
    if world == 1 and not force_dist:
      try:
        U2, p2 = lq.Gaugefields(lat), lq.Gaugefields(lat)
        lq.substitute_U_(U2, U)
        lq.gauss_distribution_(p2, 4242)
        nsw, dtau, steps = 10, 0.05, 20
        for _ in range(steps * nsw):
            eps = dtau / nsw
            lq.U_update_(U2, p2, 0.5 * eps)
            lq.P_update_(U2, p2, eps, 5.7)
            lq.U_update_(U2, p2, 0.5 * eps)
        D2 = lq.Dirac_operator(U2, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": (1, 1, 1, -1)})
        ms_d2 = lq.bench_dslash(D2, y, b, warm=20, reps=args.dslash_reps)
        msi2 = lq.bench_cg(D2, x, b, warm=5, niter=50)
        out["after_md_trajectory"] = {"link_updates": 2 * steps * nsw, "md_reunitarize": lat.get_param("md_reunitarize"),
                                      "max_unitarity_deviation": lq.unitarity_deviation(U2), "gauge_recon_active": lat.get_param("recon_active"),
                                      "plaquette": lq.calculate_Plaquette(U2), "dslash_ms": ms_d2, "cg_iters_per_s": 1e3 / msi2}
        p2.close(); U2.close()
      except Exception as e:
        out["after_md_trajectory"] = {"error": str(e)}

    # ---- secondary (outside the timed region): ONE molecular-dynamics step of the 2-flavour Wilson HMC on this lattice, measured end to end with everything resident --
    # the reference's runMD_QPQ_sw! (standardMD.jl:146-166, Sexton-Weingarten N = 10: 10 momentum updates, 20 link half steps, one P_update_fermion! = calc_UdSfdU! with
    # eps_CG 1e-16 + Traceless_antihermitian_add!) on a copy of the links, steps of 1e-9 so that the configuration stays where it is; fp64 and with the fp32 inner chain
    if world == 1 and not force_dist:
      try:
        U3, p3, G3 = lq.Gaugefields(lat), lq.Gaugefields(lat), lq.Gaugefields(lat)
        lq.substitute_U_(U3, U)
        lq.gauss_distribution_(p3, 4243)
        D3 = lq.Dirac_operator(U3, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-16})
        fa3 = lq.FermiAction(D3)
        eta3 = lq.Fermionfields(lat, lq.WILSON)
        lq.sample_pseudofermions_(eta3, U3, fa3, b)

        def md_step():
            for half in range(2):
                for _ in range(5):
                    lq.U_update_(U3, p3, 0.5e-9)
                    lq.P_update_(U3, p3, 1e-9, 5.7)
                    lq.U_update_(U3, p3, 0.5e-9)
                if half == 0:
                    lq.calc_UdSfdU_(G3, fa3, U3, eta3)
                    lq.Traceless_antihermitian_add_(p3, 1e-9, G3)
        md = {}
        for mixed in (0, 1):
            lat.set_param("mixed_action_solver", mixed)
            md_step(); device_sync()
            t0 = time.perf_counter()
            md_step(); md_step(); device_sync()      # (until round 6 a plaquette evaluation closed the window: 0.43 ms per step that no MD step contains)
            md["mixed_precision_solver_ms" if mixed else "fp64_ms"] = 1e3 * (time.perf_counter() - t0) / 2
        lat.set_param("mixed_action_solver", 0)
        md["lazy_merge"] = lat.get_param("lazy_merge")
        out["md_step_wilson_hmc"] = md
        for o in (eta3, fa3, D3, G3, p3, U3):
            o.close()
      except Exception as e:
        out["md_step_wilson_hmc"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(lq, U, b, gL)
        # max |HIP - oracle| / max |oracle| of D b (default and 18-real kernel instances) and D^+ b on the bench lattice itself
        out["fullsize_dslash_rel_err"] = out["cpu_baseline"].pop("fullsize_dslash_rel_err", None)
        out["reference_parity"] = reference_parity(lq)
    # The JSON line must be the LAST thing on the job's stdout: RCCL writes its version banner through C stdio at communicator
    # creation, where it would sit in the C buffer until exit -- every rank flushes C stdio right after comm_init and again here.
    _flush_c_stdio()
    barrier()
    if dist is not None:
        dist.destroy_process_group()
    _flush_c_stdio()
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def pmc_child(args):
    """Child of measure_traffic(): a few applications of the Wilson operator with the default links (12-real when unitary) and with all
    18 reals, nothing else -- the process rocprofv3 counts."""
    import ctypes
    import latticeqcd_jl_amd as lq
    gL = tuple(int(v) for v in args.lattice.split(","))
    lat = lq.Lattice(gL)
    for kv in args.set:
        k, v = kv.split("=")
        lat.set_param(k, int(v))
    U = lq.Gaugefields(lat)
    lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": (1, 1, 1, -1)})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    y = b.similar()
    recon0 = lat.get_param("gauge_recon")
    for recon in (recon0, 18):
        lat.set_param("gauge_recon", recon)
        for _ in range(6):
            lq.mul_(y, D, b)
    lat.set_param("gauge_recon", recon0)
    # ... and a window of the fused CG iteration (cg_roofline.traffic): D, D^+ in update mode, the two deferred-x update kernels
    x = b.similar()
    sess = lq.CGSession(D, x, b)
    sess.iterate(12)
    sess.close()
    lat.sync()


def measure_traffic(args):
    """(bytes per launch of the default Dslash kernel, the same for the all-18-reals kernel, source string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, None, "unavailable: rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="lqcd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):       # separate passes: the two do not fit the TCC counter slots together
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--lattice", args.lattice] + sum((["--set", kv] for kv in args.set), [])
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=180)
            if r.returncode != 0:
                return None, None, "unavailable: rocprofv3 --pmc %s exited %d" % (counter, r.returncode)
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and ("wilson_" in row["Kernel_Name"] or "cg_update" in row["Kernel_Name"]):
                        a = acc.setdefault(row["Kernel_Name"], [0.0, 0])
                        a[0] += float(row["Counter_Value"]); a[1] += 1
            vals[counter] = {k: v[0] / v[1] for k, v in acc.items()}
    except Exception as e:                                   # secondary: never costs the bench line
        return None, None, "unavailable: %s" % e
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def per_kernel(match):
        ks = [k for k in vals.get("FETCH_SIZE", {}) if match(k) and k in vals.get("WRITE_SIZE", {})]
        if not ks:
            return None
        k = ks[0]
        return (2.0 * vals["FETCH_SIZE"][k] + vals["WRITE_SIZE"][k]) * 1024.0
    norm = lambda k: k.replace("(bool)0", "false").replace("(bool)1", "true")
    is18 = lambda k: "wilson_" in k and "<false, false" in norm(k)
    isdag = lambda k: "wilson_" in k and "<true," in norm(k)
    t18 = per_kernel(is18)
    tdef = per_kernel(lambda k: "wilson_" in k and not is18(k) and not isdag(k)) or t18
    # per launch of the other kernels of a CG iteration (the D^+ average holds one plain launch of the set-up among its 13: -1.5 %)
    CG_TRAFFIC.clear()
    CG_TRAFFIC.update({"D": tdef, "Ddag_update_mode": per_kernel(lambda k: isdag(k) and not "<true, false" in norm(k)),
                       "cg_update_even": per_kernel(lambda k: "cg_update_even" in k), "cg_update_odd": per_kernel(lambda k: "cg_update_odd" in k)})
    src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) over 6 launches of the kernel; "
           "(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 64 B per 128-B request)")
    return tdef, t18, src


def kernel_name(lat, recon_active):
    """name of the Dslash kernel the library launches under the current tunables (what a rocprofv3 kernel trace of this run shows)"""
    v = lat.get_param("dslash_variant")
    if v != 1:
        return KERNEL_NAMES.get(v, "wilson")
    pipe = lat.get_param("dslash_pipe")
    if pipe == 1:
        return "wilson_dirsplit_pipe<false,%s,true>" % ("true" if recon_active else "false")
    if pipe == 2 and recon_active:
        return "wilson_dirsplit_s<false,true,true>"
    return "wilson_dirsplit<false,true,false>" if recon_active else "wilson_dirsplit<false,false,false>"


# |D b|^2 of the bench's synthetic problem (hot-start links seed 111, Gaussian source seed 112, kappa 0.141139, BC (1,1,1,-1)) measured on ONE GPU: the generators are
# keyed by GLOBAL site, so any decomposition of the lattice must reproduce the number -- a partitioned run that does not has a broken halo path (scripts/norm_probe.py)
DB_NORM2_ONE_GPU = {(32, 32, 32, 64): 6.63734359571458697e+07, (16, 16, 16, 32): 4.15102466603872832e+06}


def halo_selfcheck(lq, D, b, y, gL):
    """{"norm2_Db", "expected", "rel_diff", "ok"}: the partitioned operator against the one-GPU value of the same synthetic problem (None if the lattice has no entry)."""
    exp = DB_NORM2_ONE_GPU.get(tuple(gL))
    if exp is None:
        return None
    lq.mul_(y, D, b)
    got = lq.dot(y, y).real
    rel = abs(got - exp) / exp
    return {"norm2_Db": got, "expected": exp, "rel_diff": rel, "ok": bool(rel < 1e-11)}


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (the GPU boxes show 256 CPUs and grant 16:
    /sys/fs/cgroup/cpu.max = "1600000 100000" -- 256 OpenMP threads on a 16-core quota ran the all-cores leg at 2 x one core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is None:
                q = float(txt)
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                quota = None if q <= 0 else q / per
            else:
                quota = parse(txt)
            if quota:
                n = max(1, min(n, int(quota + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(lq, U, b, gL):
    """The CPU path timed on this box's host cores on a bounded sample of the same workload.  Preferred:
    the reference itself, if `julia` and its packages happen to be installed here (probed at run time, there is no network) --
    scripts/ref_cpu_baseline.jl, kind "reference".  Otherwise (the expected case): the oracle, a port of the reference algorithm,
    CG windows of 1 and 0 iterations on the SAME configuration; their difference is one full CG iteration (2 Dslash + BLAS-1), on
    ONE thread like the reference's serial Julia loop (kind "port")."""
    import shutil
    import subprocess
    probe = "julia not found on PATH"
    jl = shutil.which("julia")
    if jl:
        try:
            r = subprocess.run([jl, os.path.join(ROOT, "scripts", "ref_cpu_baseline.jl"), *[str(v) for v in gL], str(KAPPA)],
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                ref = json.loads(line[-1])
                return {"value": ref["cg_iter_per_s"], "unit": "iter/s", "cores": 1, "kind": "reference",
                        "sample": "LatticeDiracOperators.jl on the host: %s" % ref.get("sample", ""), "dslash_gflops": ref.get("dslash_gflops"),
                        "julia_probe": "julia + packages found, reference timed"}
            probe = "julia found but the reference packages did not run (exit %d): %s" % (r.returncode, (r.stderr or "").strip()[-200:])
        except Exception as e:
            probe = "julia found but the reference run failed: %s" % e
    from oracle import oracle as orc
    Uh, bh = U.download(), b.download()
    bc = (1, 1, 1, -1)
    orc.set_threads(1)
    samples = []
    for _ in range(3):      # three independent differences; `value` is their median (one difference is one sample)
        t0 = time.perf_counter(); orc.cg_DdagD_fixed(orc.WILSON, Uh, bh, gL, KAPPA, 1.0, bc, niter=0); t_setup = time.perf_counter() - t0
        t0 = time.perf_counter(); orc.cg_DdagD_fixed(orc.WILSON, Uh, bh, gL, KAPPA, 1.0, bc, niter=1); t_one = time.perf_counter() - t0
        samples.append(max(t_one - t_setup, 1e-9))
    per_iter = sorted(samples)[1]
    t0 = time.perf_counter(); ref_D = orc.wilson_D(Uh, bh, gL, KAPPA, 1.0, bc); t_d = time.perf_counter() - t0
    # the oracle's D b at the FULL bench lattice is also the parity check of the kernels this line times: the default (12-real when the links are
    # unitary) and the all-18-reals instance, and D^+ (all host cores for that one)
    parity = {}
    try:
        lat = U.lattice
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": bc})
        y = b.similar()
        scale = float(abs(ref_D).max())
        recon0 = lat.get_param("gauge_recon")
        for recon in (recon0, 18):
            lat.set_param("gauge_recon", recon)
            lq.mul_(y, D, b)
            parity["recon%d_active%d" % (recon, lat.get_param("recon_active"))] = float(abs(y.download() - ref_D).max() / scale)
        lat.set_param("gauge_recon", recon0)
        orc.set_threads(usable_cores())
        ref_Dd = orc.wilson_D(Uh, bh, gL, KAPPA, 1.0, bc, dagger=True)
        orc.set_threads(1)
        lq.mul_(y, D.adjoint(), b)
        parity["dagger"] = float(abs(y.download() - ref_Dd).max() / float(abs(ref_Dd).max()))
        del ref_Dd
        y.close()
    except Exception as e:                                   # secondary: never costs the bench line
        parity["error"] = str(e)
    del ref_D
    # the same window with the oracle's OpenMP loops on every host core, reported beside the single-thread figure (the reference's loop is
    # serial: `value` stays the 1-thread number, this one says what the box's cores could do with the same arithmetic)
    ncores = usable_cores()
    allc = None
    if ncores > 1:
        orc.set_threads(ncores)
        # NUMA-honest: the inputs are copied so that every thread first-touches the sites it owns in the threaded loops, the vectors the CG allocates
        # are first touched the same way, and the BLAS-1 loops and inner products run on all threads too (round 5: only the stencil did: 3.2 x one core on 256)
        Un, bn = orc.numa_copy(Uh, 4), orc.numa_copy(bh, 4)
        t0 = time.perf_counter(); orc.cg_DdagD_fixed(orc.WILSON, Un, bn, gL, KAPPA, 1.0, bc, niter=0); ta0 = time.perf_counter() - t0
        t0 = time.perf_counter(); orc.cg_DdagD_fixed(orc.WILSON, Un, bn, gL, KAPPA, 1.0, bc, niter=4); ta2 = time.perf_counter() - t0
        orc.wilson_D(Un, bn, gL, KAPPA, 1.0, bc)
        t0 = time.perf_counter(); orc.wilson_D(Un, bn, gL, KAPPA, 1.0, bc); ta_d = time.perf_counter() - t0
        orc.set_threads(1)
        del Un, bn
        allc = {"cores": ncores, "cpus_visible": os.cpu_count(), "value": 4.0 / max(ta2 - ta0, 1e-9), "unit": "iter/s",
                "dslash_gflops": WILSON_FLOP_PER_SITE * gL[0] * gL[1] * gL[2] * gL[3] / ta_d / 1e9,
                "sample": "the same oracle window with every loop (stencil over (t,z,y) rows, BLAS-1, inner products) on all host cores, inputs and work vectors first "
                          "touched by the owning threads: (time(4 iterations) - time(0)) / 4; a restatement of the reference's serial algorithm, not the reference"}
    return {"value": 1.0 / per_iter, "unit": "iter/s", "cores": 1, "kind": "port",
            "sample": "oracle CG on the same %dx%dx%dx%d configuration: median of 3 x [time(1 iteration) - time(0 iterations)], 1 thread" % gL,
            "samples_iter_per_s": [1.0 / t for t in samples],
            "dslash_gflops": WILSON_FLOP_PER_SITE * gL[0] * gL[1] * gL[2] * gL[3] / t_d / 1e9, "all_cores": allc, "julia_probe": probe,
            "fullsize_dslash_rel_err": parity}


def reference_parity(lq):
    """Operator-level parity against the REFERENCE ITSELF when `julia` and its packages are installed on this box (probed at run time):
    scripts/ref_parity_dump.jl writes mul!(y,D,x), mul!(y,D',x) and solve_DinvX! of LatticeDiracOperators.jl on the reference's 4^4
    fixtures with a closed-form source, the HIP path is compared with them (tests/ref_vectors.py).  Reported, never timed."""
    import shutil
    import subprocess
    import tempfile
    jl = shutil.which("julia")
    if not jl:
        return {"status": "julia not found on PATH: reference vectors absent (operator-level parity stays pinned by the oracle only)"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_vectors as rv
    out = tempfile.mkdtemp(prefix="lqcd_refvec_")
    try:
        r = subprocess.run([jl, os.path.join(ROOT, "scripts", "ref_parity_dump.jl"), out, rv.GOLDEN], capture_output=True, text=True, timeout=900)
        if r.returncode != 0 or not rv.available(out):
            return {"status": "julia found but scripts/ref_parity_dump.jl did not run (exit %d): %s" % (r.returncode, (r.stderr or "").strip()[-200:])}
        res = {"status": "reference vectors produced on this box"}
        for kind in ("wilson", "staggered"):
            lat = lq.Lattice(rv.L)
            U = lq.Gaugefields(lat).upload(lq.gauge_io.load_ildg(os.path.join(rv.GOLDEN, rv.FIXTURE[kind]), rv.L))
            D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson" if kind == "wilson" else "Staggered", "κ": rv.KAPPA, "mass": rv.MASS,
                                            "boundarycondition": rv.BC, "eps_CG": 1e-19, "MaxCGstep": 3000})
            x = lq.Fermionfields(lat, lq.WILSON if kind == "wilson" else lq.STAGGERED).upload(rv.closed_form_source(kind))
            y = x.similar()
            for op, which in ((D, "D"), (D.adjoint(), "Ddag")):
                lq.mul_(y, op, x)
                ref = rv.load(kind, which, out)
                res["%s_%s_rel_err" % (kind, which)] = float(abs(y.download() - ref).max() / abs(ref).max())
            lq.solve_DinvX_(y, lq.DdagD_operator(D), x)
            ref = rv.load(kind, "cg_x", out)
            res["%s_cg_solution_rel_err" % kind] = float(abs(y.download() - ref).max() / abs(ref).max())
        return res
    except Exception as e:
        return {"status": "reference parity run failed: %s" % e}


if __name__ == "__main__":
    main()
