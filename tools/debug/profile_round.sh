#!/bin/bash
# The three profiling passes behind a profiles/rNN?_* set (run on the GPU box through gpurun): kernel trace + stats, then the two
# PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs), each on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-configs --no-second-leg --ffn sparse
# --no-hbm-micro`.  usage: tools/debug/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/{stats,fetch,write}, gpurun_out/prof_<tag>/*.log
set -e -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--steps 2 --warmup 1 --no-cpu-baseline --ffn sparse --no-hbm-micro --no-parity --no-configs --no-second-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $FLAGS > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $FLAGS > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $FLAGS > $OUT/write.log 2>&1
echo "write pass done"
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > $OUT/pmc.log 2>&1
for f in $(find $OUT/stats -name "*kernel_trace.csv"); do python3 $R/tools/debug/gpu_idle_gaps.py $f 900 > $OUT/gpu_idle_gaps.txt 2>&1 || true; done
# keep the merged-back files small: the per-dispatch traces are large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT $OUT/stats/* | head -40
