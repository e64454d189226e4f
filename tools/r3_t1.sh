cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
MMDA_BENCH_BACKEND=gloo MMDA_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --global-stats 1 --no-cpu-baseline > gpurun_out/bench_g2gs.json 2> gpurun_out/bench_g2gs.err; echo "g2 global stats rc=$?"; cut -c1-200 gpurun_out/bench_g2gs.json; tail -3 gpurun_out/bench_g2gs.err
MMDA_BENCH_BACKEND=gloo MMDA_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "g2 rc=$?"; cut -c1-200 gpurun_out/bench_g2.json
