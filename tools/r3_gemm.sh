cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_gemm_r3.py 32 256 > gpurun_out/gemm_dma2.log 2>&1 || { echo FAILED dma2; tail -20 gpurun_out/gemm_dma2.log; exit 1; }
MMDA_GEMM_DMA_STAGES=3 timeout -k 10 300 python tools/bench_gemm_r3.py 32 256 > gpurun_out/gemm_dma3.log 2>&1 || { echo FAILED dma3; tail -20 gpurun_out/gemm_dma3.log; exit 1; }
MMDA_GEMM_DMA=0 timeout -k 10 300 python tools/bench_gemm_r3.py 32 256 > gpurun_out/gemm_old.log 2>&1 || { echo FAILED old; tail -20 gpurun_out/gemm_old.log; exit 1; }
MMDA_GEMM_DMA=0 MMDA_GEMM_TN=0 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm_old.log 2>&1; tail -3 gpurun_out/t_gemm_old.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > gpurun_out/t_gemm.log 2>&1; tail -3 gpurun_out/t_gemm.log
cat gpurun_out/gemm_dma2.log
