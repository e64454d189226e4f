# soak: long runs of the fused step (flag joins, roles, early sort) -- any device-side wait that times out or any recurrence abort fails the bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "32 20000" "64 10000" "256 4000" "16 10000"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --batch $1 --steps $2 --warmup 20 --no-cpu-baseline > gpurun_out/soak_b$1.json 2> gpurun_out/soak_b$1.err; echo "B=$1 steps=$2 rc=$? $(cut -c100-180 gpurun_out/soak_b$1.json)"
done
timeout -k 10 400 python bench.py --batch 32 --steps 5000 --warmup 20 --ragged 1 --confidnet 1 --no-cpu-baseline > gpurun_out/soak_rc.json 2> gpurun_out/soak_rc.err; echo "ragged+confid rc=$? $(cut -c100-180 gpurun_out/soak_rc.json)"
