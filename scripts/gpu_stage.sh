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
  suite)      # what the driver does at round end
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3
    timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 2>&1 | tail -25 | tee $out/pytest.log
    timeout 900 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 2500 $out/bench_n1.json; tail -3 $out/bench_n1.err
    ;;
  *) echo "unknown stage $stage"; exit 1;;
esac
