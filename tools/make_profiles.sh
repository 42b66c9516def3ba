#!/bin/bash
# Collects this round's profiling evidence on the GPU box into gpurun_out/prof/ (run through gpurun from the
# repository root).  Every GPU step has its own time limit; the chain stops at the first failure.
# tools/collect_profiles.py then condenses the outputs into profiles/.
#   1. kernel trace + stats of the bench command itself (its kernel averages are what bench.py's HIP-event times must agree with)
#      and once more with --no-hd, so that the per-kernel averages of the stats file are 4K launches only
#   2. counters over tools/prof_4k.py (the 4K legs of bench.py, one picture), one rocprofv3 run per pass:
#      SQ instruction / wait counters, FETCH_SIZE, WRITE_SIZE
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/stats" -o bench --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/stats_4k" -o bench4k --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --no-hd > "$O/bench_4k_under_rocprof.json" 2> "$O/bench_4k_under_rocprof.err" &&
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$O/pmc_sq" -o sq --output-format csv -- python3 "$R/tools/prof_4k.py" > "$O/pmc_sq.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d "$O/pmc_fetch" -o fetch --output-format csv -- python3 "$R/tools/prof_4k.py" > "$O/pmc_fetch.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d "$O/pmc_write" -o write --output-format csv -- python3 "$R/tools/prof_4k.py" > "$O/pmc_write.log" 2>&1 &&
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d "$O/stats4k" -o p4k --output-format csv -- python3 "$R/tools/prof_4k.py" 5 > "$O/stats4k.log" 2>&1 &&
cd "$R" &&
timeout -k 10 500 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" &&
echo PROFILES_OK
