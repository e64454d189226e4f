# end-of-round measurements: configuration sweep, then the profiles of B = 32 and B = 256
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/r3_sweep.sh 2>&1 | tee gpurun_out/sweep.txt
bash tools/r3_profiles.sh 2>&1 | tail -5
