#!/bin/bash
# bash profiles/pmc.sh <tag> "<counter set 1>" "<counter set 2>" ...   (one rocprofv3 --pmc pass per set,
# on `bench.py --main-only`: only the headline forward launches are in the trace)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_set$i -- python3 $REPO/bench.py --main-only --steps 5 --warmup 1 ${BENCH_ARGS} > /dev/null 2> $OUT/${TAG}_pmc_set$i.err || echo "set $i ($C) failed"
done
