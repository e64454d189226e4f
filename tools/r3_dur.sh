cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --durations=40 > gpurun_out/t_dur.log 2>&1; echo "rc=$?" >> gpurun_out/t_dur.log
grep -A45 "slowest" gpurun_out/t_dur.log | head -50
