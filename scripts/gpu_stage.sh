#!/bin/bash
# One script for every GPU lease of a round: `gpurun -- bash scripts/gpu_stage.sh <stage>`; results under gpurun_out/<stage>/.
cd "$(dirname "$0")/.."
stage=${1:-suite}
out=gpurun_out/$stage
mkdir -p $out
export TMPDIR=/tmp
case $stage in
  rhmc)       # the action handle, the full-size oracle parity tests and config 3's stopping rule
    timeout 1500 python -m pytest tests/test_gpu_rhmc.py tests/test_gpu_rational.py tests/test_gpu_fullsize.py tests/test_gpu_md_staggered.py tests/test_gpu_md_mixed.py \
        tests/test_gpu_mixed.py tests/test_gpu_reference_callers.py -q -x --durations=15 2>&1 | tail -40 > $out/pytest.log
    timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "evenodd or c_abi or bicgstab or fermi or force" 2>&1 | tail -15 >> $out/pytest.log
    tail -25 $out/pytest.log
    timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 1500 $out/bench_n1.json
    ;;
  lazy)       # lazy link triples below the C ABI, the one-launch CG's give-up protocol
    timeout 1500 python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_md.py tests/test_gpu_md_partitioned.py tests/test_gpu_md_staggered.py tests/test_gpu_reunit.py \
        tests/test_gpu_cg_persist.py tests/test_gpu_lifecycle.py tests/test_gpu_hmc_partitioned.py tests/test_gpu_solver_edges.py tests/test_gpu_graph.py -q -x --durations=8 2>&1 | tail -30 > $out/pytest.log
    timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "c_abi or cg" 2>&1 | tail -15 >> $out/pytest.log
    cat $out/pytest.log
    ;;
  bicg)       # fused even-odd BiCGStab chain, action solves through it; config timings
    timeout 900 python -m pytest tests/test_gpu_solver_edges.py tests/test_gpu_clover.py tests/test_gpu_md.py tests/test_gpu_hmc_partitioned.py tests/test_gpu_fullsize.py -q 2>&1 | tail -15 > $out/pytest.log
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cg_persist.py tests/test_gpu_graph.py tests/test_gpu_halo_fuse.py -q 2>&1 | tail -15 >> $out/pytest.log
    cat $out/pytest.log
    timeout 900 python scripts/bench_configs.py > $out/bench_configs.log 2> $out/bench_configs.err; cut -c1-1500 $out/bench_configs.log; tail -5 $out/bench_configs.err
    ;;
  bicgprof)   # kernel traces of the even-odd BiCGStab chain in its forms + first differing iteration
    python scripts/bicg_probe.py --repeat --L 8,8,8,16 2>&1 | tail -18 | tee $out/diff.log
    python scripts/bicg_probe.py 0 1 2 2>&1 | tail -4 | tee $out/times.log
    for m in 0 2; do
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$m -o p -- python $GRAFT_REPO_ROOT/scripts/bicg_probe.py $m --reps 10 2>&1 | tail -3)
      f=$(find $out/prof$m -name "*kernel_stats.csv" | head -1); echo "== mode $m ($f)"; ls -R $out/prof$m | head -20
      if [ -n "$f" ]; then head -14 "$f" | cut -d, -f1-6 | sed 's/"//g' | cut -c1-200; cp "$f" $out/kernel_stats_mode$m.csv; fi
    done
    ;;
  sweep18)    # workgroup-map / cache-hint sweep of the all-18-reals Dslash (the kernel reference-format configurations take)
    for set in "xcd_nsub=16 xcd_ysplit=4" "xcd_nsub=32 xcd_ysplit=4" "xcd_nsub=32 xcd_ysplit=8" "xcd_nsub=16 xcd_ysplit=2" "xcd_nsub=8 xcd_ysplit=2" "xcd_nsub=64 xcd_ysplit=8" \
               "nt_gauge=0" "nt_gauge=3" "nt_store=0" "dslash_block=64"; do
      args=""; for kv in $set; do args="$args --set $kv"; done
      python scripts/dslash_probe.py --reps 200 --warm 20 --set gauge_recon=18 $args 2>&1 | tail -1
    done | tee $out/sweep18.log
    ;;
  round)      # full GPU suite, then the round's rocprofv3 evidence (scripts/gpu_profile_round.sh) and the configuration timings
    timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -20 | tee $out/pytest.log
    bash scripts/gpu_profile_round.sh ${2:-r04} 2>&1 | tail -5
    timeout 900 python scripts/bench_configs.py > $out/bench_configs.log 2> $out/bench_configs.err; tail -3 $out/bench_configs.log | cut -c1-600
    ;;
  cloverprof) # kernel stats of the even-odd Wilson-clover BiCGStab at 32^3x64, fused and generic chain
    for m in 0 2; do
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$m -o p -- python $GRAFT_REPO_ROOT/scripts/bicg_probe.py $m --reps 3 --L 32,32,32,64 --csw 1.0 2>&1 | grep bicg_fused)
      f=$(find $out/prof$m -name "*kernel_stats.csv" | head -1); echo "== mode $m"; head -9 "$f" | cut -d, -f1-4 | sed 's/"//g' | cut -c1-150
    done
    ;;
  delta)      # "12 + delta" links for reference-format configurations: timing and parity
    python scripts/delta_probe.py 2>&1 | tail -5 | tee $out/delta.log
    python scripts/delta_probe.py --L 16,16,16,32 --dev 1e-9 2>&1 | tail -4 | tee -a $out/delta.log
    timeout 600 python -m pytest tests/test_gpu_recon12.py tests/test_gpu_pipe.py tests/test_golden_io.py tests/test_gpu_parity.py -q -x 2>&1 | tail -8 | tee $out/pytest.log
    timeout 600 python bench.py --no-pmc > $out/bench.json 2> $out/bench.err; python -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['roofline']['frac'], d['gauge_recon18_all_reals_read'], d['reference_format_links'])"
    ;;
  mixedeo)    # mixed-precision even-odd BiCGStab
    timeout 900 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_md_mixed.py tests/test_gpu_pair32.py -q -x 2>&1 | tail -12 | tee $out/pytest.log
    timeout 900 python scripts/bench_configs.py > $out/bench_configs.log 2> $out/bench_configs.err; sed -n 5p $out/bench_configs.log | cut -c800-1900; sed -n 6p $out/bench_configs.log; tail -3 $out/bench_configs.err
    ;;
  merge)      # waiting link / momentum updates (lazy_merge), measured MD step, bench.py after the settle change
    timeout 1200 python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_md.py tests/test_gpu_md_partitioned.py tests/test_gpu_md_staggered.py tests/test_gpu_md_mixed.py \
        tests/test_gpu_reunit.py tests/test_gpu_hmc_partitioned.py tests/test_gpu_lifecycle.py tests/test_gpu_clover.py tests/test_gpu_domainwall.py -q -x --durations=5 2>&1 | tail -14 | tee $out/pytest.log
    timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "c_abi" 2>&1 | tail -3 | tee -a $out/pytest.log
    timeout 900 python scripts/bench_configs.py > $out/bench_configs.log 2> $out/bench_configs.err; sed -n 5p $out/bench_configs.log | cut -c1-900; tail -3 $out/bench_configs.err
    timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; python -c "import json; d=json.load(open('$out/bench.json')); print(d['value'], d['dslash_ms'], d['roofline']['frac'], d['gauge_recon18_all_reals_read']['dslash_ms'], d['reference_format_links'])"; tail -2 $out/bench.err
    ;;
  pu)         # momentum + link update as one sweep vs two passes
    timeout 600 python -m pytest tests/test_gpu_md.py tests/test_gpu_reunit.py tests/test_gpu_reference_callers.py tests/test_gpu_stout.py -q -x 2>&1 | tail -3
    python scripts/pu_probe.py 0.005 2>&1 | tail -1 | tee $out/pu.log
    python scripts/pu_probe.py 0.05 2>&1 | tail -1 | tee -a $out/pu.log
    if [ -f latticeqcd.jl_amd/csrc/liblqcd_hip_ab.so ]; then LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip_ab.so python scripts/pu_probe.py 0.005 2>&1 | tail -1 | sed 's/^/A-B build (LQCD_EXTRA_FLAGS of the day): /' | tee -a $out/pu.log; fi
    ;;
  forceprof)  # kernel stats of calc_UdSfdU! at 32^3x64, fp64 even-odd and mixed
    for m in 0 1; do
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$m -o p -- python $GRAFT_REPO_ROOT/scripts/force_probe.py $m 3 2>&1 | tail -2)
      f=$(find $out/prof$m -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_mixed$m.csv
      python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:22]:
    print("%-90s %6s calls %9.3f ms %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
    done
    ;;
  dw)         # Domainwall operator, action, force, the reference's test case
    timeout 900 python -m pytest tests/test_gpu_domainwall.py -q -x --durations=6 2>&1 | tail -25 | tee $out/pytest.log
    ;;
  stout)      # stout smearing layer, back-propagation, the callers' CovNeuralnet path
    timeout 900 python -m pytest tests/test_gpu_stout.py -q -x --durations=6 2>&1 | tail -25 | tee $out/pytest.log
    ;;
  mdprof)     # kernel stats of the MD step (one-sweep momentum + link update), the stout layer and the Domainwall operator
    for m in 0 1; do
      (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$m -o p -- python $GRAFT_REPO_ROOT/scripts/md_probe.py $m 2>&1 | grep "MD step")
      f=$(find $out/prof$m -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_mixed$m.csv
      python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:26]:
    print("%-100s %6s calls %9.3f ms avg %8.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
    done
    ;;
  stapleab)   # A/B builds and map settings of the staple sweep (P_update! / one-sweep momentum + link update)
    for lib in "" _ab_OCC1 _ab_OCC3 _ab_BURST0; do
      f=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip$lib.so; [ -f $f ] || continue
      LQCD_HIP_LIB=$f python scripts/pu_probe.py 0.005 2>&1 | tail -1 | sed "s/^/lib$lib: /" | tee -a $out/stapleab.log
    done
    for set in "md_remap=0" "xcd_nsub=8" "xcd_nsub=32" "xcd_ysplit=2" "xcd_ysplit=8" "staple_recon=0"; do
      python scripts/pu_probe.py 0.005 $set 2>&1 | tail -1 | tee -a $out/stapleab.log
    done
    ;;
  mappmc)     # fabric read traffic (FETCH_SIZE) and time of the default Dslash under workgroup-map settings
    for set in "xcd_nsub=16 xcd_ysplit=4" "xcd_nsub=8 xcd_ysplit=2" "xcd_nsub=8 xcd_ysplit=4" "xcd_nsub=16 xcd_ysplit=2" "xcd_nsub=4 xcd_ysplit=2" "xcd_nsub=32 xcd_ysplit=4"; do
      args=""; for kv in $set; do args="$args --set $kv"; done
      t=$(python scripts/dslash_probe.py --reps 200 --warm 20 $args 2>&1 | tail -1)
      (cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc -o p -- python $GRAFT_REPO_ROOT/scripts/dslash_probe.py --reps 5 --warm 1 $args > /dev/null 2>&1)
      f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
      python - "$f" "$set" "$t" <<'PY' | tee -a $out/mappmc.log
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wilson_dirsplit" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = [float(r["Counter_Value"]) for r in rows]
fetch = sum(v) / len(v)
tot = (2 * fetch + 393216.0) * 1024
print("%-28s FETCH_SIZE %.4e KiB -> traffic %.3f GB = %.3f x 768 B/site | %s" % (sys.argv[2], fetch, tot / 1e9, tot / (768 * 2097152), sys.argv[3][:160]))
PY
      rm -rf $out/pmc
    done
    ;;
  pupmc)      # counters of the one-sweep momentum + link update kernel (separate --pmc passes, kernel trace only)
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
      n=$(echo $pass | cut -d' ' -f1)
      (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$n -o p -- python $GRAFT_REPO_ROOT/scripts/pu_probe.py 0.005 > /dev/null 2>&1)
      f=$(find $out/pmc_$n -name "*counter_collection.csv" | head -1)
      python - "$f" <<'PY' | tee -a $out/pupmc.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gauge_force_kernel" in k or "link_exp_update" in k:
        acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-22s mean %.5e over %d launches" % (k, c, sum(v) / len(v), len(v)))
PY
    done
    ;;
  fold)       # round 5: folded halo schedule (boundary hops inside the stencil launch) -- tests, the N = 8 / 4 / 2 one-die proxy with halo_fold 0 / 1, kernel timelines
    timeout 900 python -m pytest tests/test_gpu_halo_fuse.py -q -x 2>&1 | tail -6 | tee $out/pytest.log
    timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "half_lattice or rccl or c_abi or partitioned" 2>&1 | tail -4 | tee -a $out/pytest.log
    timeout 900 python -m pytest tests/test_gpu_rhmc.py tests/test_gpu_reference_callers.py tests/test_gpu_md_partitioned.py tests/test_gpu_hmc_partitioned.py -q -x 2>&1 | tail -4 | tee -a $out/pytest.log
    for f in 0 1; do for extra in "" "--set nt_blas=0" "--set nt_blas=0 --set nt_store=0"; do
      LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 --set halo_stream_mode=3 --set halo_fold=$f $extra 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 fold=$f /"; echo
    done; done | tee $out/proxy.log
    for f in 0 1; do
      LQCD_FORCE_PARTITION=12 timeout 200 python scripts/dslash_probe.py --lattice 32,32,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 300 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=4 fold=$f /"; echo
      LQCD_FORCE_PARTITION=8 timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,32 --selfcomm 1 --reps 200 --warm 20 --cg 300 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=2 fold=$f /"; echo
    done | tee -a $out/proxy.log
    LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 tuner /" | tee -a $out/proxy.log; echo
    timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 200 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=1 /" | tee -a $out/proxy.log; echo
    for f in 0 1; do
      (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace$f -o t -- python $GRAFT_REPO_ROOT/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 64 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep "^cg")
      t=$(find $out/trace$f -name "*kernel_trace.csv" | head -1)
      echo "== halo_fold $f (halo_stream_mode 3)" | tee -a $out/timeline.log
      python scripts/timeline.py "$t" cg_update_odd 2>&1 | tee -a $out/timeline.log
      rm -rf $out/trace$f
    done
    ;;
  call2)      # round 5, call 2: the whole GPU suite on the tree with the trace-replayed callers, fused pack + reduction, DW5 without spills; proxy; bench line
    timeout 1800 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -30 | tee $out/pytest.log
    for f in 0 1; do
      LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 fold=$f /"; echo
    done | tee $out/proxy.log
    LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 tuner /" | tee -a $out/proxy.log; echo
    timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 200 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=1 /" | tee -a $out/proxy.log; echo
    (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace1 -o t -- python $GRAFT_REPO_ROOT/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 64 --set halo_stream_mode=3 --set halo_fold=1 2>&1 | grep "^cg")
    t=$(find $out/trace1 -name "*kernel_trace.csv" | head -1)
    echo "== halo_fold 1 (halo_stream_mode 3), pack + reduction in one launch" | tee $out/timeline.log
    python scripts/timeline.py "$t" cg_update_odd 2>&1 | tee -a $out/timeline.log
    rm -rf $out/trace1
    timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 3000 $out/bench_n1.json; tail -3 $out/bench_n1.err
    LQCD_BENCH_FORCE_DIST=1 LQCD_FORCE_PARTITION=14 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --lattice 32,16,16,32 --no-cpu-baseline --no-pmc > $out/bench_proxy_n8local.json 2> $out/bench_proxy.err; python -c "
import json; d=json.load(open('$out/bench_proxy_n8local.json')); print({k: d.get(k) for k in ('value','ms_per_step','halo_phases_ms_max_over_ranks','halo_stream_mode_rank0','allreduce_latency_us')})"; tail -3 $out/bench_proxy.err
    ;;
  call3)      # round 5, call 3: where did the suite stall (per-test timeout with stack dump), the ring of search-direction buffers (cg_defer_x = K), proxy with it
    timeout 900 python -m pytest tests/test_gpu_reference_callers.py -q -x --timeout=200 --timeout-method=thread --durations=12 2>&1 | tail -60 | tee $out/pytest_callers.log
    timeout 600 python -m pytest tests/test_gpu_solver_edges.py tests/test_gpu_stout.py -q -x --timeout=200 --timeout-method=thread --durations=6 2>&1 | tail -25 | tee $out/pytest_edges.log
    for k in 2 4 8; do
      timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 100 --warm 20 --cg 300 --set cg_defer_x=$k 2>&1 | grep -E "^cg" | sed "s/^/N=1 cg_defer_x=$k /"
      LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 50 --warm 20 --cg 400 --set halo_stream_mode=3 --set cg_defer_x=$k 2>&1 | grep -E "^cg" | sed "s/^/N=8 fold=1 cg_defer_x=$k /"
    done | tee $out/ring.log
    ;;
  call4)      # round 5, call 4: RCCL point-to-point channel settings against the halo exchange of the N = 8 proxy; the replayed callers with the oracle on one thread
    for e in "" "NCCL_NCHANNELS_PER_PEER=4" "NCCL_NCHANNELS_PER_PEER=8" "NCCL_NCHANNELS_PER_PEER=16" "NCCL_NCHANNELS_PER_PEER=32" "NCCL_MIN_P2P_NCHANNELS=8 NCCL_MAX_P2P_NCHANNELS=8" "NCCL_MIN_P2P_NCHANNELS=16 NCCL_MAX_P2P_NCHANNELS=16" "NCCL_MIN_NCHANNELS=32" "NCCL_P2P_NET_CHUNKSIZE=1048576 NCCL_NCHANNELS_PER_PEER=8"; do
      env $e LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 --set halo_stream_mode=3 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/[$e] /"; echo
    done | tee $out/nccl_env.log
    (cd /tmp && LQCD_FORCE_PARTITION=14 NCCL_NCHANNELS_PER_PEER=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace1 -o t -- python $GRAFT_REPO_ROOT/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 64 --set halo_stream_mode=3 2>&1 | grep "^cg")
    t=$(find $out/trace1 -name "*kernel_trace.csv" | head -1)
    echo "== NCCL_NCHANNELS_PER_PEER=8" | tee $out/timeline.log
    python scripts/timeline.py "$t" cg_update_ring 2>&1 | tee -a $out/timeline.log
    rm -rf $out/trace1
    timeout 600 python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_solver_edges.py -q -x --timeout=200 --timeout-method=thread --durations=5 2>&1 | tail -12 | tee $out/pytest.log
    ;;
  call5)      # round 5, call 5: the folded schedule for every operator (clover, staggered, fp32, x-partitioned): every self-partition test, proxies of configs 4 and 5
    timeout 1500 python -m pytest tests/test_gpu_halo_fuse.py tests/test_gpu_clover.py tests/test_gpu_mixed.py tests/test_gpu_md_partitioned.py tests/test_gpu_hmc_partitioned.py \
        tests/test_gpu_bench_dist.py tests/test_gpu_stout.py tests/test_gpu_solver_edges.py -q -x --timeout=300 --timeout-method=thread --durations=8 2>&1 | tail -22 | tee $out/pytest_a.log
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domainwall.py tests/test_gpu_pipe.py -q -x -k "rccl or partition" --timeout=300 --timeout-method=thread 2>&1 | tail -6 | tee $out/pytest_b.log
    for f in 0 1; do
      LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --kind Staggered --lattice 48,24,24,48 --selfcomm 1 --reps 100 --warm 20 --cg 200 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/staggered 48^3x96 N=8 local fold=$f /"; echo
      LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --kind WilsonClover --lattice 32,16,16,32 --selfcomm 1 --reps 100 --warm 20 --cg 200 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/clover 32^3x64 N=8 local fold=$f /"; echo
      LQCD_FORCE_PARTITION=15 timeout 200 python scripts/dslash_probe.py --lattice 16,16,16,32 --selfcomm 1 --reps 100 --warm 20 --cg 200 --set halo_stream_mode=3 --set halo_fold=$f 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/wilson N=16 local (x too) fold=$f /"; echo
    done | tee $out/proxy_configs45.log
    LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 100 --warm 20 --cg 400 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/wilson N=8 tuner /" | tee -a $out/proxy_configs45.log; echo
    timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 100 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=1 /" | tee -a $out/proxy_configs45.log; echo
    ;;
  proxyprof)  # round 5: rocprofv3 evidence of the partitioned CG at the N = 8 local volume (folded schedule): kernel stats, fabric traffic of its kernels
    P="python $GRAFT_REPO_ROOT/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 400 --set halo_stream_mode=3"
    (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -o p -- $P 2>&1 | grep "^cg")
    f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv; head -12 $out/kernel_stats.csv | cut -c1-200
    for ctr in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$ctr -o p -- python $GRAFT_REPO_ROOT/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 5 --warm 2 --cg 40 --set halo_stream_mode=3 > /dev/null 2>&1)
    done
    python - $out <<'PY' | tee $out/pmc_traffic.log
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/pmc_" + ctr + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]][ctr].append(float(r["Counter_Value"]))
print("# fabric traffic per launch at the N = 8 local volume 32x16x16x32 (262144 sites): (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md")
for k, v in sorted(acc.items()):
    if v["FETCH_SIZE"] and v["WRITE_SIZE"]:
        fe, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        b = (2 * fe + wr) * 1024
        print("%-72s launches %4d  traffic %8.2f MB = %6.0f B/site" % (k, len(v["FETCH_SIZE"]), b / 1e6, b / 262144))
PY
    rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
    ;;
  suite)      # what the driver does at round end
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3
    timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 2>&1 | tail -25 | tee $out/pytest.log
    timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 2500 $out/bench_n1.json; tail -3 $out/bench_n1.err
    ;;
  *) echo "unknown stage $stage"; exit 1;;
esac
