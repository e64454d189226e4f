# round-3 profiles: kernel-trace stats + three PMC passes of the bench at B = 32 and B = 256 (CSV left under gpurun_out/ for tools/summarize_profiles.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for B in 32 256; do
  bash tools/prof_bench.sh $B > /dev/null 2>&1
  bash tools/prof_pmc.sh $B > /dev/null 2>&1
  python tools/trace_step.py $(ls gpurun_out/prof_b$B/*/*kernel_trace.csv | head -1) > gpurun_out/trace_b$B.txt 2>&1
done
du -sh gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cut -c1-300 gpurun_out/bench_driver_style.json
