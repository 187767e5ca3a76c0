#!/bin/bash
# Kernel trace + PMC passes of the Wilson Dslash.  usage: gpu_profile.sh <tag> [probe args...]
cd "$(dirname "$0")/../.."
R=$(pwd); TAG=$1; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -o t -- python $R/scripts/dslash_probe.py --reps 50 --cg 20 "$@" > $R/gpurun_out/$TAG/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/$TAG/pmc_$name -o p -- python $R/scripts/dslash_probe.py --reps 5 --warm 1 "$@" > $R/gpurun_out/$TAG/pmc_$name.log 2>&1
  echo "pass [$pass] rc=$?"
done
cd $R
find gpurun_out/$TAG -name "*.csv" | head -30
python scripts/summarize_prof.py gpurun_out/$TAG 2>&1 | tail -40
