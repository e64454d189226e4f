cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "base:" "nodw:MMDA_DW_OVERLAP=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for B in 64 128; do
    env $envs timeout -k 10 300 python bench.py --batch $B --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab_${name}_b${B}_$rep.json 2> gpurun_out/ab_${name}_b${B}_$rep.err
    echo "$name B=$B rep=$rep $(python -c "import json,sys; d=json.load(open('gpurun_out/ab_${name}_b${B}_$rep.json')); print(d['ms_per_step'], d['roofline']['all_launch_ms'])")"
  done
done
done
