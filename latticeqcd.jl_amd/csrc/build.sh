#!/bin/bash
# Builds liblqcd_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   LQCD_EXTRA_FLAGS="-DLQCD_GAUGE_AOSOA=0" LQCD_OUT=liblqcd_hip_soa.so ./build.sh   builds an A/B variant
set -e
cd "$(dirname "$0")"
ARCH=gfx950      # MI355X only: cg_persist.hip's 72 KiB of static LDS, the MFMA-free fp64 kernels and every tuning constant assume CDNA4
OUT=${LQCD_OUT:-liblqcd_hip.so}
BDIR=build/${OUT%.so}
# LQCD_VARIANTS=1: also build the measured-and-slower Wilson kernel variants 2-8 (experiments/stencil_alt/stencil_alt.hip, dslash_variant >= 2) -- an experiment
# build (experiments/stencil_alt/README.md); the product library carries the default kernels only (a dslash_variant >= 2 then runs variant 1)
ALT=""; ALTO=""
if [ "${LQCD_VARIANTS:-0}" = "1" ]; then LQCD_EXTRA_FLAGS="$LQCD_EXTRA_FLAGS -DLQCD_VARIANTS"; ALT="stencil_alt"; BDIR=${BDIR}_variants; OUT=${LQCD_OUT:-liblqcd_hip_variants.so}; fi
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-unused-variable $LQCD_EXTRA_FLAGS"
mkdir -p $BDIR
pids=()
if [ -n "$ALT" ]; then
  mkdir -p $BDIR
  if [ ! -f $BDIR/stencil_alt.o ] || [ ../../experiments/stencil_alt/stencil_alt.hip -nt $BDIR/stencil_alt.o ] || [ lqcd_internal.h -nt $BDIR/stencil_alt.o ] || [ stencil_common.h -nt $BDIR/stencil_alt.o ]; then
    ( hipcc $FLAGS -I. -c ../../experiments/stencil_alt/stencil_alt.hip -o $BDIR/stencil_alt.o ) &
    pids+=($!)
  fi
fi
for f in stencil stencil_pair32 cg_persist fields blas apply comm solvers actions rational bench_api mdom capi force mixed md clover domainwall; do
  if [ ! -f $BDIR/$f.o ] || [ $f.hip -nt $BDIR/$f.o ] || [ lqcd_internal.h -nt $BDIR/$f.o ] || [ ops_internal.h -nt $BDIR/$f.o ] || [ stencil_common.h -nt $BDIR/$f.o ] || [ ../../include/lqcd_hip.h -nt $BDIR/$f.o ] || [ build.sh -nt $BDIR/$f.o ]; then
    # the compiler's per-kernel resource remarks (registers, scratch, occupancy) are kept beside the object: scripts/resource_usage.py, tests/test_kernel_resources.py
    ( set +e; hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $f.hip -o $BDIR/$f.o 2> $BDIR/$f.remarks; rc=$?; if [ $rc != 0 ]; then grep -v "remark:" $BDIR/$f.remarks >&2 || true; else grep -A3 "warning:" $BDIR/$f.remarks >&2 || true; fi; exit $rc ) &
    pids+=($!)
  fi
done
if [ ! -f $BDIR/stencil32.o ] || [ stencil.hip -nt $BDIR/stencil32.o ] || [ stencil_common.h -nt $BDIR/stencil32.o ] || [ lqcd_internal.h -nt $BDIR/stencil32.o ] || [ build.sh -nt $BDIR/stencil32.o ]; then
  ( set +e; hipcc $FLAGS -DLQCD_F32 -Rpass-analysis=kernel-resource-usage -c stencil.hip -o $BDIR/stencil32.o 2> $BDIR/stencil32.remarks; rc=$?; if [ $rc != 0 ]; then grep -v "remark:" $BDIR/stencil32.remarks >&2 || true; else grep -A3 "warning:" $BDIR/stencil32.remarks >&2 || true; fi; exit $rc ) &   # fp32 build of the stencil (inner solver of the mixed-precision CG)
  pids+=($!)
fi
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
[ $fail = 0 ] || { echo "compilation failed" >&2; exit 1; }
hipcc --offload-arch=$ARCH -shared -fPIC -o $OUT $BDIR/stencil.o $BDIR/stencil32.o ${ALT:+$BDIR/stencil_alt.o} $BDIR/stencil_pair32.o $BDIR/cg_persist.o $BDIR/fields.o $BDIR/blas.o $BDIR/apply.o $BDIR/comm.o $BDIR/solvers.o $BDIR/actions.o $BDIR/rational.o $BDIR/bench_api.o $BDIR/mdom.o $BDIR/capi.o $BDIR/force.o $BDIR/mixed.o $BDIR/md.o $BDIR/clover.o $BDIR/domainwall.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $(pwd)/$OUT"
