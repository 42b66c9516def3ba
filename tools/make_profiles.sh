#!/bin/bash
# Collects this round's profiling evidence on the GPU box into gpurun_out/prof/ (run through gpurun from the
# repository root).  Every GPU step has its own time limit; the chain stops at the first failure.
# tools/collect_profiles.py then condenses the outputs into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 "$R/tools/ubench_valu" > "$O/ubench_valu.txt" 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/stats" -o bench --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --no-4k > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err" &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$O/pmc_sq" -o sq --output-format csv -- python3 "$R/tools/prof_me.py" all > "$O/pmc_sq.log" 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$O/pmc_fetch" -o fetch --output-format csv -- python3 "$R/tools/prof_me.py" all > "$O/pmc_fetch.log" 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$O/pmc_write" -o write --output-format csv -- python3 "$R/tools/prof_me.py" all > "$O/pmc_write.log" 2>&1 &&
cd "$R" &&
timeout -k 10 500 python3 tests/tools/report_configs.py > "$O/configs.json" 2> "$O/configs.err" &&
timeout -k 10 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" &&
echo PROFILES_OK
