# PMC passes of the bench (program directly after `--`; counters in passes of their own, kernel-trace only beside them):
#   FETCH_SIZE, WRITE_SIZE            HBM-side traffic per dispatch (MI355X guide, section HBM: FETCH_SIZE x2 for wide streams)
#   SQ_VALU_MFMA_BUSY_CYCLES & co     MFMA-busy / SQ-busy (north_star: "rocprof HBM GB/s and MFMA-busy counters")
# usage: bash tools/prof_pmc.sh [batch]     (default 32 = the headline config; 256 = the per-GPU batch of BASELINE configs[2])
# Counter collection serialises the dispatches of the process: a kernel that waits on the device for another stream's kernel (the flag
# joins of the fused step) would wait for a launch that cannot start -- the step falls back to event joins for these passes.
export MMDA_FLAG_JOIN=0 MMDA_SORT_EARLY=0
B=${1:-32}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch_b$B gpurun_out/pmc_write_b$B gpurun_out/pmc_mfma_b$B
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_b$B -- python bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/pmc_fetch_b$B.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_b$B -- python bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/pmc_write_b$B.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_b$B -- python bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/pmc_mfma_b$B.log 2>&1
ls gpurun_out/pmc_fetch_b$B/*/ gpurun_out/pmc_write_b$B/*/ gpurun_out/pmc_mfma_b$B/*/
