#!/bin/bash
# counter passes over tools/prof_refine.py (argv: PU size, 0 = one list of all sizes; picture): issue-side and memory-side
# counters of refine_kernel and cost_kernel in separate rocprofv3 runs; output under gpurun_out/prof_refine/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_refine
rm -rf "$O"; mkdir -p "$O"
S=${1:-16}; P=${2:-4k}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY -d "$O/sq" -o sq --output-format csv -- python3 "$R/tools/prof_refine.py" "$S" "$P" > "$O/sq.log" 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY -d "$O/ta" -o ta --output-format csv -- python3 "$R/tools/prof_refine.py" "$S" "$P" > "$O/ta.log" 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH -d "$O/sq2" -o sq2 --output-format csv -- python3 "$R/tools/prof_refine.py" "$S" "$P" > "$O/sq2.log" 2>&1 &&
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "x265hip::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("x265hip::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k[0]:16s} {k[1]:40s} {sum(acc[k]) / len(acc[k]):16.0f}  ({len(acc[k])} launches)")
PY
