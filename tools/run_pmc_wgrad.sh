# PMC study of the fp32 weight-gradient kernel on the head shape: usage tools/run_pmc_wgrad.sh TAG [dtype]
TAG=$1; DT=${2:-f32}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python /root/repo/tools/kbench.py --which wgrad --dtype $DT --reps 2"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o run -- $CMD > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
done
cd /root/repo
python tools/pmc_kernel.py $OUT/p1 $OUT/p2 $OUT/p3 --match conv_wgrad > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
