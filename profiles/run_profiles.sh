#!/bin/bash
# Run on the GPU box through gpurun:  bash profiles/run_profiles.sh <tag>
# Produces gpurun_out/<tag>_*; `python profiles/summarize.py <tag>` condenses them into profiles/.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench done" >> $OUT/${TAG}_progress.log
# (a) headline kernel alone: the --stats average IS the per-launch duration of the timed launches
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_statsmain -- python3 $REPO/bench.py --main-only --steps 20 --warmup 3 > $OUT/${TAG}_statsmain_bench.json 2> $OUT/${TAG}_statsmain.err || echo "statsmain failed" >> $OUT/${TAG}_bench.err
echo "statsmain done" >> $OUT/${TAG}_progress.log
# (b) the whole bench (forward, adjoint, f32, single-timestep launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py --no-cpu --steps 20 --warmup 3 > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err || echo "stats failed" >> $OUT/${TAG}_bench.err
echo "stats done" >> $OUT/${TAG}_progress.log
# (c) PMC passes, one counter set per run (TCC: FETCH_SIZE needs 3 of 4 slots)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$N -- python3 $REPO/bench.py --no-cpu --steps 5 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_$N.err || echo "pmc $C failed" >> $OUT/${TAG}_bench.err
  echo "pmc $N done" >> $OUT/${TAG}_progress.log
done
