# kernel-trace + stats of the bench; usage: bash tools/prof_bench.sh [batch]   (default 32)
B=${1:-32}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b$B
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$B -- python bench.py --steps 30 --warmup 5 --batch $B --no-cpu-baseline > gpurun_out/prof_b$B.log 2>&1
ls -R gpurun_out/prof_b$B | head -20
