#!/bin/bash
# counters of me_team_kernel on the 8x8 / 16x16 calls of one 4K picture x 1 reference (tools/prof_me_variants.py), one pass per run
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_team
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for s in ${SIZES:-8 16}; do
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$O/sq$s" -o sq --output-format csv -- python3 "$R/tools/prof_me_variants.py" $s > "$O/sq$s.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$O/sqb$s" -o sq --output-format csv -- python3 "$R/tools/prof_me_variants.py" $s > "$O/sqb$s.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU -d "$O/sqc$s" -o sq --output-format csv -- python3 "$R/tools/prof_me_variants.py" $s > "$O/sqc$s.log" 2>&1 || exit 1
done
python3 - <<PY
import csv, collections
for s in "${SIZES:-8 16}".split():
    for d in ('sq', 'sqb', 'sqc'):
        rows = list(csv.DictReader(open(f'$O/{d}{s}/sq_counter_collection.csv')))
        by = collections.OrderedDict()
        for r in rows:
            if 'team' not in r['Kernel_Name']: continue
            k = int(r['Dispatch_Id'])
            by.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
            by[k]['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        for k, v in list(by.items())[:1]:
            print(s, d, ' '.join(f'{a}={b:.4g}' for a, b in v.items()))
PY
echo TEAM_PROF_OK
