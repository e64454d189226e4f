cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof1.log 2>&1
ls -R gpurun_out/prof1 | head -20
