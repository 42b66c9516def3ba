#!/bin/bash
# refine_kernel with and without the XCD-banded placement of its batches (X265_RF_BANDS): FETCH_SIZE per launch over
# tools/prof_4k.py and the kernel's time in the same driver
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_bands
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for b in 0 1; do
export X265_RF_BANDS=$b
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d "$O/fetch$b" -o fetch --output-format csv -- python3 "$R/tools/prof_4k.py" > "$O/fetch$b.log" 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d "$O/stats$b" -o st --output-format csv -- python3 "$R/tools/prof_4k.py" 5 > "$O/stats$b.log" 2>&1 || exit 1
done
python3 - <<PY
import csv, collections, glob
for b in (0, 1):
    rows = list(csv.DictReader(open(glob.glob(f'$O/fetch{b}/**/*counter_collection.csv', recursive=True)[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r['Kernel_Name'].split('(')[0][-40:]].append(float(r['Counter_Value']))
    st = list(csv.DictReader(open(glob.glob(f'$O/stats{b}/**/*kernel_stats.csv', recursive=True)[0])))
    print('bands', b)
    for k, v in acc.items():
        t = [s for s in st if s['Name'].split('(')[0][-40:] == k]
        print(f'  {k:42s} FETCH_SIZE {sum(v) / len(v) * 1024 / 1e6:9.1f} MB/launch (raw)  avg {float(t[0]["AverageNs"]) / 1e3 if t else -1:9.1f} us')
PY
echo BANDS_OK
