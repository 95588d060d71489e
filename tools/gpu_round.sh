#!/bin/bash
# One GPU-box visit: parity tests, the bench line, rocprofv3 kernel stats and the PMC passes.  usage: tools/gpu_round.sh TAG [steps...]
# Every step has its own timeout and failure does not stop the next one; everything lands under gpurun_out/TAG/.
TAG=${1:-r02}; shift
STEPS=${*:-"tests bench stats pmc"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
BENCH_TRAIN="python bench.py --no-cpu-baseline --no-inference --no-parity-mode"
for s in $STEPS; do
  case $s in
    tests)  timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "tests rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/pytest.log ;;
    bench)  timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.txt; tail -c 3000 $OUT/bench.log ;;
    stats)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/stats_bf16 -o run -- python /root/repo/bench.py --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 10 --warmup 3 > /root/repo/$OUT/stats_bf16.log 2>&1); echo "stats_bf16 rc=$?" | tee -a $OUT/rc.txt
            (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/stats_f32 -o run -- python /root/repo/bench.py --dtype f32 --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 4 --warmup 2 > /root/repo/$OUT/stats_f32.log 2>&1); echo "stats_f32 rc=$?" | tee -a $OUT/rc.txt ;;
    pmc)    for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
              n=$(echo $p | cut -d' ' -f1)
              (cd /tmp && timeout 600 rocprofv3 --pmc $p --output-format csv -d /root/repo/$OUT/pmc_$n -o run -- python /root/repo/bench.py --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 2 --warmup 1 > /root/repo/$OUT/pmc_$n.log 2>&1); echo "pmc $n rc=$?" | tee -a $OUT/rc.txt
            done
            python tools/pmc_summary.py $OUT/pmc_bf16.json bf16 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $OUT/pmc_summary.log 2>&1; cat $OUT/pmc_summary.log
            # the raw counter CSVs are large: keep the summaries only
            rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES ;;
    pmcf32) (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /root/repo/$OUT/pmcf_SQ -o run -- python /root/repo/bench.py --dtype f32 --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 1 --warmup 1 > /root/repo/$OUT/pmcf_SQ.log 2>&1); echo "pmcf32 rc=$?" | tee -a $OUT/rc.txt
            python tools/pmc_summary.py $OUT/pmc_f32.json f32 $OUT/none $OUT/none $OUT/pmcf_SQ > $OUT/pmcf_summary.log 2>&1; cat $OUT/pmcf_summary.log; rm -rf $OUT/pmcf_SQ ;;
    pmcmodes) for dt in f32 f32_bf16x3; do
              for pp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
                n=$(echo $pp | cut -d' ' -f1)
                (cd /tmp && timeout 600 rocprofv3 --pmc $pp --output-format csv -d /root/repo/$OUT/pm_${dt}_$n -o run -- python /root/repo/bench.py --dtype $dt --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 1 --warmup 1 > /root/repo/$OUT/pm_${dt}_$n.log 2>&1); echo "pmc $dt $n rc=$?" | tee -a $OUT/rc.txt
              done
              python tools/pmc_summary.py $OUT/pmc_$dt.json $dt $OUT/pm_${dt}_FETCH_SIZE $OUT/pm_${dt}_WRITE_SIZE $OUT/pm_${dt}_SQ_VALU_MFMA_BUSY_CYCLES > $OUT/pmc_${dt}_summary.log 2>&1; cat $OUT/pmc_${dt}_summary.log
              rm -rf $OUT/pm_${dt}_FETCH_SIZE $OUT/pm_${dt}_WRITE_SIZE $OUT/pm_${dt}_SQ_VALU_MFMA_BUSY_CYCLES
            done ;;
    pmcx3)  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /root/repo/$OUT/pmcx_SQ -o run -- python /root/repo/bench.py --dtype f32_bf16x3 --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 1 --warmup 1 > /root/repo/$OUT/pmcx_SQ.log 2>&1); echo "pmcx3 rc=$?" | tee -a $OUT/rc.txt
            python tools/pmc_summary.py $OUT/pmc_bf16x3.json f32_bf16x3 $OUT/none $OUT/none $OUT/pmcx_SQ > $OUT/pmcx_summary.log 2>&1; cat $OUT/pmcx_summary.log; rm -rf $OUT/pmcx_SQ ;;
    statsx3) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/stats_x3 -o run -- python /root/repo/bench.py --dtype f32_bf16x3 --no-cpu-baseline --no-inference --no-parity-mode --no-roofline --steps 6 --warmup 2 > /root/repo/$OUT/stats_x3.log 2>&1); echo "stats_x3 rc=$?" | tee -a $OUT/rc.txt ;;
    statsinf) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/stats_infer -o run -- python /root/repo/tools/infer_bench.py --network efficientdet-d0 --batch 32 --size 512 --reps 5 > /root/repo/$OUT/stats_infer.log 2>&1); echo "stats_infer rc=$?" | tee -a $OUT/rc.txt ;;
    bw)     hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_bw tools/probe/stream_bw.hip > /dev/null 2>&1 && timeout 120 /tmp/stream_bw > $OUT/stream_bw.txt 2>&1; cat $OUT/stream_bw.txt
            hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probe/mfma_peak.hip > /dev/null 2>&1 && timeout 120 /tmp/mfma_peak > $OUT/mfma_peak.txt 2>&1; cat $OUT/mfma_peak.txt ;;
    *)      echo "unknown step $s" ;;
  esac
done
# trim the kernel-trace dumps to the stats tables
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
ls -la $OUT
