#!/bin/bash
# round 3, GPU call K: hop order of the scalar-addressing kernel (both uses of a link in the same phase of concurrent workgroups)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_k; rm -rf $O; mkdir -p $O gpurun_out/r03
for rep in 1 2; do
for recon in 12 18; do
  for ho in 0 1 2; do
    for ntg in 1 0; do
      timeout 100 python scripts/dslash_probe.py --reps 200 --warm 20 --cg 100 --set dslash_pipe=4 --set gauge_recon=$recon --set hop_order=$ho --set nt_gauge=$ntg 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/recon$recon hop_order=$ho nt_gauge=$ntg /"; echo
    done
  done
  timeout 100 python scripts/dslash_probe.py --reps 200 --warm 20 --cg 100 --set dslash_pipe=0 --set gauge_recon=$recon 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/recon$recon variant1 /"; echo
done; done | tee $O/hop_order.log
bash scripts/r03/pmc_ab.sh ho1_12 dslash_pipe=4 hop_order=1 > $O/pmc1.log 2>&1
bash scripts/r03/pmc_ab.sh ho1_18 dslash_pipe=4 hop_order=1 gauge_recon=18 > $O/pmc2.log 2>&1
bash scripts/r03/pmc_ab.sh ho0_18 dslash_pipe=4 hop_order=0 gauge_recon=18 > $O/pmc3.log 2>&1
grep "FETCH\|TCC_" gpurun_out/r03/pmc_ho1_12.csv gpurun_out/r03/pmc_ho1_18.csv gpurun_out/r03/pmc_ho0_18.csv
