cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_dp.py -q -m gpu -x > gpurun_out/t_sel.log 2>&1; echo "rc=$?" >> gpurun_out/t_sel.log
tail -6 gpurun_out/t_sel.log
MMDA_BENCH_BACKEND=gloo MMDA_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "g2 rc=$?"; cut -c1-250 gpurun_out/bench_g2.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
