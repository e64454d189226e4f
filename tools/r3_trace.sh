# kernel timeline of one steady-state step at B=32 and B=256 (tools/trace_step.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 32 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$B -- python bench.py --steps 30 --warmup 5 --batch $B --no-cpu-baseline > gpurun_out/prof_b$B.log 2>&1
python tools/trace_step.py $(ls gpurun_out/prof_b$B/*/*kernel_trace.csv | head -1) > gpurun_out/trace_b$B.txt 2>&1
cp $(ls gpurun_out/prof_b$B/*/*kernel_stats.csv | head -1) gpurun_out/stats_b$B.csv
rm -rf gpurun_out/prof_b$B
done
