cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
tail -5 gpurun_out/t_all.log
for B in 32 256; do
  timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err
  cut -c100-215 gpurun_out/bench_b$B.json
done
