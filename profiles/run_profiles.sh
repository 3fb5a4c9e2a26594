#!/bin/bash
# Run on the GPU box through gpurun:  bash profiles/run_profiles.sh <tag> [part]
#   part 1: bench line + rocprofv3 --kernel-trace --stats of every leg           (~5 min)
#   part 2: PMC passes of the forward kernel (+ the float32 fast mode), one counter set per run   (~6 min)
#   part 3: PMC passes of the adjoint and the tricubic forward                   (~5 min)
#   part 5: PMC passes of the Fermat kernels                                     (~2 min)
# Produces gpurun_out/<tag>_*; `python profiles/summarize.py <tag>` condenses them into profiles/.
# (--pmc runs carry --kernel-trace only: gpurun refuses PMC combined with other trace domains.)
TAG=${1:-r02}
PART=${2:-1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
note() { echo "$(date +%T) $1" >> $OUT/${TAG}_progress.log; }

stats() {   # stats <name> <bench args...>
  local N=$1; shift
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_$N -- python3 $REPO/bench.py "$@" > $OUT/${TAG}_stats_$N.json 2> $OUT/${TAG}_stats_$N.err || note "stats $N FAILED"
  note "stats $N done"
}
pmc() {     # pmc <leg> <set index> "<counters>"
  local LEG=$1 I=$2 C=$3
  # (a counter set the hardware cannot schedule aborts the profiled process, which can then hang in finalisation:
  #  bound every run)
  timeout -k 5 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${LEG}_$I -- python3 $REPO/bench.py --only $LEG --steps 5 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_${LEG}_$I.err || note "pmc $LEG set $I ($C) FAILED"
  # keep only this library's kernels (the torch helper kernels of the workload set-up are most of the rows; gpurun copies back <= 64 MiB)
  for f in $OUT/${TAG}_pmc_${LEG}_$I/*/*_counter_collection.csv; do
    [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep "::k_" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
  done
  rm -f $OUT/${TAG}_pmc_${LEG}_$I/*/*_kernel_trace.csv
  note "pmc $LEG $I done"
}

if [ "$PART" = "1" ]; then
  python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
  note "bench done"
  # (--settle-ms 150 as in the default bench run: ~1 000 untimed launches of the leg first, so the average is the SUSTAINED per-launch
  #  duration the bench line times -- bench.py:settle, profiles/r05_clock_ramp.json)
  stats forward --only forward --steps 20 --warmup 3 --settle-ms 600
  stats f32_forward --only f32_forward --steps 20 --warmup 3 --settle-ms 600
  stats adjoint --only adjoint --steps 10 --warmup 2 --settle-ms 150
  stats cubic_forward --only cubic_forward --steps 10 --warmup 2 --settle-ms 150
  stats cubic_adjoint --only cubic_adjoint --steps 10 --warmup 2 --settle-ms 150
  stats cgls --only cgls --steps 30 --warmup 1 --settle-ms 150
  stats sirt --only sirt --steps 30 --warmup 1 --settle-ms 150
  stats all --no-cpu --steps 20 --warmup 3
fi
SETS=("FETCH_SIZE" "WRITE_SIZE"
      "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
      "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_DRAM_sum"
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
      "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
      "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
      "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
      "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS")
FIRST=${3:-1}
if [ "$PART" = "2" ]; then
  i=0; for C in "${SETS[@]}"; do i=$((i+1)); [ $i -ge $FIRST ] && pmc forward $i "$C"; done
  for i in 1 3 5 6 7 8 10; do pmc f32_forward $i "${SETS[$((i-1))]}"; done      # the float32 fast mode (k_forward_bundle_f32)
fi
if [ "$PART" = "4" ]; then      # selected sets for one leg:  run_profiles.sh <tag> 4 <leg> "<set indices>"
  for i in $4; do pmc $3 $i "${SETS[$((i-1))]}"; done
fi
if [ "$PART" = "3" ]; then
  i=0; for C in "${SETS[@]}"; do i=$((i+1)); [ $i -ge $FIRST ] && pmc adjoint $i "$C"; done
  for i in 1 3 4 5 6 7 8 10; do pmc cubic_forward $i "${SETS[$((i-1))]}"; done
  for i in 4 5 7 8 10; do pmc cubic_adjoint $i "${SETS[$((i-1))]}"; done
fi
if [ "$PART" = "5" ]; then      # the Fermat integrator: 620 000 curved rays (both indices) and config 3
  for i in 5 6 7 8; do pmc fermat_cubic $i "${SETS[$((i-1))]}"; done
  for i in 5 6 7 8; do pmc fermat_linear $i "${SETS[$((i-1))]}"; done
  for i in 7 8; do pmc fermat_cfg3 $i "${SETS[$((i-1))]}"; done
fi
