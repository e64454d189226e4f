cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
tail -4 gpurun_out/t_all.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err
cut -c1-200 gpurun_out/bench_driver_style.json
