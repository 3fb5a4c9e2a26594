#!/bin/bash
# Run on the GPU box through gpurun:  bash profiles/run_profiles.sh <tag>
# Produces gpurun_out/<tag>_{bench.json,stats,pmc_*}; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py --no-cpu --steps 20 --warmup 3 > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err || exit 2
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$N -- python3 $REPO/bench.py --no-cpu --steps 5 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_$N.err || echo "pmc $C failed" >> $OUT/${TAG}_bench.err
done
find $OUT -name "*_kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
