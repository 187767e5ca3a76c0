cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bicg_probe.py 4 --L 32,32,32,64 --reps 3 --csw 1.0 2>&1 | tail -1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
head -8 $f | sed 's/(lqcd::[A-Za-z:0-9]*Args)//' | cut -c1-150
