cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_gemm_r3.py 32 256 > gpurun_out/gemm_dma2b.log 2>&1 || { echo FAILED dma2; tail -20 gpurun_out/gemm_dma2b.log; exit 1; }
grep "bwd L\|====" gpurun_out/gemm_dma2b.log
for B in 32 256; do
  timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err || { echo FAILED bench $B; tail -5 gpurun_out/bench_b$B.err; exit 1; }
  MMDA_GEMM_DMA=0 timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b${B}_nodma.json 2> gpurun_out/bench_b${B}_nodma.err
  MMDA_GEMM_DMA=0 MMDA_GEMM_TN_MAX_ROWS=4096 timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b${B}_nodma_tn4096.json 2> /dev/null
done
for f in gpurun_out/bench_b*.json; do echo $f; cut -c1-210 $f; done
