cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lm_pmc2
mkdir -p $O
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_DRAM_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/set$i -- python3 $R/profiles/tools/time_cubic_bundle.py > $O/set$i.json 2> $O/set$i.err || echo "set $i FAILED"
done
python3 $R/profiles/tools/pmc_kernel_mean.py k_forward_ $O/set1 $O/set2 $O/set3 $O/set4 > $O/summary.json
