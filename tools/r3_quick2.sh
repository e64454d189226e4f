cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x -k "gemm or linear or fp32_fused or fp32_unfused or bf16_path" > gpurun_out/t_sel.log 2>&1; echo "rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
for B in 32 256; do
  timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err
  cut -c100-215 gpurun_out/bench_b$B.json
done
timeout -k 10 300 python bench.py --precision fp32 --no-cpu-baseline --steps 50 > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err; cut -c100-215 gpurun_out/bench_fp32.json
for B in 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$B -- python bench.py --steps 30 --warmup 5 --batch $B --no-cpu-baseline > gpurun_out/prof_b$B.log 2>&1
python tools/trace_step.py $(ls gpurun_out/prof_b$B/*/*kernel_trace.csv | head -1) > gpurun_out/trace_b$B.txt 2>&1
cp $(ls gpurun_out/prof_b$B/*/*kernel_stats.csv | head -1) gpurun_out/stats_b$B.csv
rm -rf gpurun_out/prof_b$B
done
