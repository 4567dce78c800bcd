#!/bin/bash
# Kernel-trace statistics and HBM/SQ counters of `bench.py` (each counter group in a pass of its own).
# usage: tools/profile.sh <tag> <workload> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
WL=${1:-config2}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# One stream, no graph replay: per-kernel durations are then the ones bench.py's per-stage HIP events measure (with the
# forked shadow stream two kernels share the GPU and each one's own duration stretches).
export RT_BENCH_OPTS=${RT_BENCH_OPTS:-overlap=0,graphs=0}
# one frame at a time: frames in flight share the GPU and stretch every kernel's own duration
ARGS="--workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-scaling-4k --no-variants --frames-in-flight 1 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $ARGS > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py $ARGS > "$OUT/pmc_sq2.log" 2>&1 || { echo "sq2 pass failed"; tail -5 "$OUT/pmc_sq2.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc_l2" -- python3 bench.py $ARGS > "$OUT/pmc_l2.log" 2>&1 || { echo "l2 pass failed"; tail -5 "$OUT/pmc_l2.log"; exit 1; }
{ echo "library options during profiling: $RT_BENCH_OPTS   bench args: $ARGS"; python3 tools/summarize_profile.py "$OUT" "$WL"; } > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# the raw per-dispatch tables are tens of megabytes per workload; what is kept is the summary, the per-kernel statistics and counters.json
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
