# PMC study of one implicit-GEMM variant on the head-tower shape: usage tools/run_pmc.sh TAG VARIANT
TAG=$1; V=$2; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python /root/repo/tools/kbench2.py --reps 3 --variants $V --only 0"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o run -- $CMD > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
done
cd /root/repo
python tools/pmc_kernel.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 --match conv_igemm > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
