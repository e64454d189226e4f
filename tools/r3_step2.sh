cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 64 128; do
  for G in 0 1000; do
  MMDA_GEMM_DMA_MIN_GFLOP=$G timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b${B}_g$G.json 2> gpurun_out/bench_b${B}_g$G.err
  done
done
for f in gpurun_out/bench_b*_g*.json; do echo $f; cut -c100-210 $f; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b256 -- python bench.py --steps 20 --warmup 5 --batch 256 --no-cpu-baseline > gpurun_out/prof_b256.log 2>&1
python tools/trace_step.py $(ls gpurun_out/prof_b256/*/*kernel_trace.csv | head -1) > gpurun_out/trace_b256.txt 2>&1
rm -rf gpurun_out/prof_b256
cat gpurun_out/trace_b256.txt
