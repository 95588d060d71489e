#!/bin/bash
# copy the judged summaries of the last round-6 evidence call (tools/profile_round6.sh, tools/r6_list.sh, bench.py, the -m gpu suite) from gpurun_out/ to profiles/
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r06
for m in bf16 f32 f32_bf16x3 f32_bwd_bf16x3 f32_hf16x3_bwd_bf16x3; do cp $P/kt_$m/kt_kernel_stats.csv profiles/r06_bench_kernel_stats_$m.csv; done
for n in d0 d4; do cp $P/kt_infer_$n/kt_kernel_stats.csv profiles/r06_infer_kernel_stats_${n}_f32_hf16x3_bwd_bf16x3.csv; cp $P/pmc_infer_${n}_f32_hf16x3_bwd_bf16x3.json profiles/r06_pmc_infer_${n}_f32_hf16x3_bwd_bf16x3.json; done
cp $P/pmc_f32_hf16x3_bwd_bf16x3.json profiles/r06_pmc_f32_hf16x3_bwd_bf16x3.json
cp $P/step_trace_f32_hf16x3_bwd_bf16x3.txt profiles/r06_step_trace_f32_hf16x3_bwd_bf16x3.txt
cp gpurun_out/step_list.txt profiles/r06_step_list.txt
cp gpurun_out/parity_errors.txt profiles/r06_parity_errors.txt
tail -1 gpurun_out/bench_r06_final.json > profiles/r06_bench_line.json
ls -la profiles/r06_* | wc -l
