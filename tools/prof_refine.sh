set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_refine
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d "$O/a" -o a --output-format csv -- python3 "$R/tools/prof_refine.py" 16 > "$O/a.log" 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE -d "$O/b" -o b --output-format csv -- python3 "$R/tools/prof_refine.py" 16 > "$O/b.log" 2>&1 &&
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d "$O/s" -o s --output-format csv -- python3 "$R/tools/prof_refine.py" 16 > "$O/s.log" 2>&1 &&
echo PROF_OK
