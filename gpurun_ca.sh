cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ca_pmc
mkdir -p $O
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/set$i -- python3 $R/bench.py --only cubic_adjoint --steps 3 --warmup 1 > $O/set$i.json 2> $O/set$i.err || echo "set $i FAILED"
done
python3 $R/profiles/tools/pmc_kernel_mean.py k_adjoint_binned $O/set1 $O/set2 $O/set3 > $O/summary.json
python3 $R/profiles/tools/pmc_kernel_mean.py k_lm_fold $O/set1 $O/set2 $O/set3 > $O/summary_fold.json
