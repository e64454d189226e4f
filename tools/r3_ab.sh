# selected parity tests, then bench runs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "golden or train_epoch or bf16_path or c3_ or c5 or determin or shape_changes or cmd or loss or tall or one_rank" > gpurun_out/t_sel.log 2>&1; echo "rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
for rep in 1 2; do
for cfg in "base:" "zgmain:MMDA_ZERO_GRAD_SIDE=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for B in 32 64; do
    env $envs timeout -k 10 300 python bench.py --batch $B --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab_${name}_b${B}_$rep.json 2> gpurun_out/ab_${name}_b${B}_$rep.err
    echo "$name B=$B rep=$rep $(python -c "import json,sys; d=json.load(open('gpurun_out/ab_${name}_b${B}_$rep.json')); print(d['ms_per_step'])")"
  done
done
done
