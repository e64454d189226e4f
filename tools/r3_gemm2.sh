cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_gemm_r3.py 256 > gpurun_out/gemm_tall3.log 2>&1 || { echo FAILED tall3; tail -20 gpurun_out/gemm_tall3.log; exit 1; }
MMDA_GEMM_DMA_TALL_STAGES=2 timeout -k 10 300 python tools/bench_gemm_r3.py 256 > gpurun_out/gemm_tall2.log 2>&1 || { echo FAILED tall2; tail -20 gpurun_out/gemm_tall2.log; exit 1; }
MMDA_GEMM_DMA_TALL=0 timeout -k 10 300 python tools/bench_gemm_r3.py 256 > gpurun_out/gemm_notall.log 2>&1
MMDA_GEMM_DMA_MIN_ROWS=0 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1; tail -3 gpurun_out/t_gemm.log
for B in 256; do
  timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err; cut -c100-215 gpurun_out/bench_b$B.json
  MMDA_GEMM_DMA_TALL=0 timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b${B}_notall.json 2> /dev/null; cut -c100-215 gpurun_out/bench_b${B}_notall.json
  MMDA_GEMM_DMA_TALL_STAGES=2 timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b${B}_tall2.json 2> /dev/null; cut -c100-215 gpurun_out/bench_b${B}_tall2.json
done
