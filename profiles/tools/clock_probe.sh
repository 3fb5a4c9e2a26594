#!/bin/bash
# bash profiles/tools/clock_probe.sh <tag> <lib or ""> ...   -> gpurun_out/<tag>_clock_<i>/  (one rocprofv3 --pmc pass per library)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for LIB in "$@"; do
  i=$((i+1))
  if [ "$LIB" = "default" ]; then unset IONOTOMO_LIB; else export IONOTOMO_LIB=$REPO/$LIB; fi
  timeout -k 5 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv \
     -d $REPO/gpurun_out/${TAG}_clock_$i -- python3 $REPO/bench.py --only forward --steps 10 --warmup 2 > /dev/null 2> $REPO/gpurun_out/${TAG}_clock_$i.err || echo "pass $i failed"
done
python3 $REPO/profiles/tools/clock_probe.py k_forward_bundle $REPO/gpurun_out/${TAG}_clock_*/
