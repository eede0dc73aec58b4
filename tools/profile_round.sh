#!/bin/bash
# rocprofv3 evidence of one bench configuration, run on the GPU box:  tools/profile_round.sh <tag> [bench.py args...]
#   gpurun_out/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1 --no-cpu-baseline <args>`
#   gpurun_out/<tag>_bench_under_rocprof.json the bench line that run printed
#   gpurun_out/<tag>_pmc_traffic.json         three --pmc passes (one counter group each, no tracing flags) folded by tools/pmc_summary.py
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
# counter passes run for minutes without output: a ticker keeps the call from being taken for hung
( while true; do sleep 45; echo "[profile] $tag: still running ($(date +%T))"; done ) &
TICK=$!
trap "kill $TICK 2>/dev/null" EXIT
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag $R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes --host-init "$@" > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> /tmp/prof_$tag/kt.err
f=$(find /tmp/prof_$tag/kt -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
echo "[profile] kernel trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    d=/tmp/prof_$tag/pmc_$(echo $grp | cut -d' ' -f1)
    timeout -k 10 420 rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-modes --host-init "$@" > /dev/null 2> $d.err
    echo "[profile] pmc $grp done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc_traffic.json /tmp/prof_$tag/pmc_FETCH_SIZE /tmp/prof_$tag/pmc_WRITE_SIZE /tmp/prof_$tag/pmc_SQ_VALU_MFMA_BUSY_CYCLES > /dev/null
head -8 $R/gpurun_out/${tag}_kernel_stats.csv
