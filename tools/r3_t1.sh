cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "schedule_switches or tall_class" > gpurun_out/t_sel.log 2>&1; echo "rc=$?" >> gpurun_out/t_sel.log
tail -30 gpurun_out/t_sel.log
