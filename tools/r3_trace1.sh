# kernel timeline of one steady-state step at B=32 with the side stream off (what do the gaps in front of the backward recurrences belong to?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export MMDA_NO_SIDE=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ns -- python bench.py --steps 30 --warmup 5 --batch 32 --no-cpu-baseline > gpurun_out/prof_ns.log 2>&1
python tools/trace_step.py $(ls gpurun_out/prof_ns/*/*kernel_trace.csv | head -1) > gpurun_out/trace_noside.txt 2>&1
rm -rf gpurun_out/prof_ns
