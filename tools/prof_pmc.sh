# HBM traffic counters of the dominant kernels (separate passes for FETCH_SIZE and WRITE_SIZE, MI355X guide section HBM)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/
