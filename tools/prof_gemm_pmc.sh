# where the waves of the bf16 GEMM spend their cycles (B = 256 text problems, one problem per launch)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gemm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_gemm -- python tools/bench_gemm_b256.py 256 > gpurun_out/pmc_gemm.log 2>&1
ls gpurun_out/pmc_gemm/*/
