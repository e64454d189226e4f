cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gemm -- python tools/bench_gemm.py > gpurun_out/prof_gemm.log 2>&1
ls gpurun_out/prof_gemm/*/ | head
