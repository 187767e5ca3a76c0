import csv, glob, sys
from collections import defaultdict
a = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wilson' in r['Kernel_Name']:
            a[r['Counter_Name']][0] += float(r['Counter_Value']); a[r['Counter_Name']][1] += 1
m = {k: v[0] / v[1] for k, v in a.items()}
if m:
    print("hit=%.3f L2req=%.2fGB EA_RD=%.2fGB" % (m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']),
          (m['TCC_HIT_sum'] + m['TCC_MISS_sum']) * 128 / 1e9, m['TCC_EA0_RDREQ_sum'] * 128 / 1e9))
