#!/bin/bash
# which stream of which direction misses L2?  FETCH_SIZE of the Wilson split kernel with one stream redirected to an L2-hot chunk
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/ablate2; mkdir -p $O; export TMPDIR=/tmp
LIST="0 256 512 1024 2048 4096 8192 16384 32768 65536 131072 262144 524288 3840 61440 983040"
for d in $LIST; do
  (cd /tmp && timeout 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$d -o p -- python $R/scripts/dslash_probe.py --reps 4 --warm 1 --set dbg=$d > $O/f$d.log 2>&1) || echo "pmc failed"
done
python - <<PY
import csv,glob
names={0:"all",256:"spinor x hot",512:"spinor y hot",1024:"spinor z hot",2048:"spinor t hot",4096:"gauge fwd x hot",8192:"gauge fwd y hot",16384:"gauge fwd z hot",32768:"gauge fwd t hot",65536:"gauge bwd x hot",131072:"gauge bwd y hot",262144:"gauge bwd z hot",524288:"gauge bwd t hot",3840:"all neighbour spinors hot",61440:"all fwd links hot",983040:"all bwd links hot"}
base=None
for d in [int(x) for x in "$LIST".split()]:
    v=[float(r['Counter_Value']) for f in glob.glob("gpurun_out/ablate2/f%d/**/*counter_collection.csv"%d,recursive=True) for r in csv.DictReader(open(f)) if 'wilson_dirsplit' in r['Kernel_Name']]
    if not v: continue
    gb=2*1024*sum(v)/len(v)/1e9
    if base is None: base=gb
    print("%-28s read %.3f GB   saved %.3f GB"%(names[d], gb, base-gb))
PY
