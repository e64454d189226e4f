# kernel-trace stats of the B=256 (C3 per-GPU) step and of the headline B=32 step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b256 -- python bench.py --steps 20 --warmup 5 --batch 256 --no-cpu-baseline > gpurun_out/prof_b256.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof1.log 2>&1
ls gpurun_out/prof_b256/*/ | head
