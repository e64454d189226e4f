cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dp.py -q -m gpu -x > gpurun_out/t_sel.log 2>&1; echo "rc=$?" >> gpurun_out/t_sel.log
tail -40 gpurun_out/t_sel.log
