#!/bin/bash
# Run ON THE GPU BOX (via gpurun).  Produces the rocprofv3 evidence bench.py's numbers
# are checked against: kernel-trace stats of the bench command itself, and separate
# PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) for HBM traffic.
# usage: tools/collect_profiles.sh <round-tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (bench.py asks for 8 hardware queues from inside Python, but rocprofv3's preloaded library has initialised the runtime by
# then: the profiled runs get the same 8 queues only if the variable is in the environment before rocprofv3 starts)
export GPU_MAX_HW_QUEUES=8
ARGS="--steps 50 --warmup 5 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1 || exit 3
cd $ROOT
python3 tools/summarize_profiles.py $OUT $TAG
# 10 M-point frame (BASELINE configs[2]: plane + cylinder RANSAC): per-kernel durations for the streaming-kernel roofline rows
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace10m -- python3 $ROOT/tools/stage_times.py --points 10000000 --flags 12 --reps 5 > $OUT/stage10m.log 2>&1 || exit 4
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc10m_fetch -- python3 $ROOT/tools/stage_times.py --points 10000000 --flags 12 --reps 3 > $OUT/stage10m_fetch.log 2>&1 || echo "10 M FETCH_SIZE pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc10m_write -- python3 $ROOT/tools/stage_times.py --points 10000000 --flags 12 --reps 3 > $OUT/stage10m_write.log 2>&1 || echo "10 M WRITE_SIZE pass failed"
cd $ROOT
cp "$(ls $OUT/trace10m/*/*kernel_stats.csv | tail -1)" $OUT/summary/${TAG}_kernel_stats_10M.csv
python3 tools/streaming_roofline.py "$(ls $OUT/trace10m/*/*kernel_stats.csv | tail -1)" $OUT/stage10m.log $OUT/summary/${TAG}_streaming_kernels_10M.json $OUT/pmc10m_fetch $OUT/pmc10m_write
# one frame at a time (no other frame's kernels beside it): the exclusive per-kernel durations
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_one -- python3 $ROOT/tools/stage_times.py --flags 8 --reps 40 > $OUT/stage_one.log 2>&1 || exit 5
cd $ROOT
cp "$(ls $OUT/trace_one/*/*kernel_stats.csv | tail -1)" $OUT/summary/${TAG}_kernel_stats_one_frame.csv
