# round-3 baseline: new tests + timelines of the B=32 and B=256 steps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dp.py tests/test_gpu_model.py -x -q -m gpu -k "one_rank_rccl or c5_as_one" > gpurun_out/t_new.log 2>&1; echo "tests rc=$?" >> gpurun_out/t_new.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench20.json 2> gpurun_out/bench20.err
python bench.py --no-cpu-baseline > gpurun_out/bench200.json 2> gpurun_out/bench200.err
MMDA_BENCH_BACKEND=gloo MMDA_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b32 -- python bench.py --steps 30 --warmup 5 --batch 32 --no-cpu-baseline > gpurun_out/prof_b32.log 2>&1
python tools/trace_step.py $(ls gpurun_out/prof_b32/*/*kernel_trace.csv | head -1) > gpurun_out/trace_b32.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b256 -- python bench.py --steps 20 --warmup 5 --batch 256 --no-cpu-baseline > gpurun_out/prof_b256.log 2>&1
python tools/trace_step.py $(ls gpurun_out/prof_b256/*/*kernel_trace.csv | head -1) > gpurun_out/trace_b256.txt 2>&1
rm -rf gpurun_out/prof_b32 gpurun_out/prof_b256
tail -3 gpurun_out/t_new.log; cat gpurun_out/bench20.json gpurun_out/bench200.json gpurun_out/bench_g2.json | cut -c1-400
