# A/B of one environment switch of DESIGN.md section 7a: alternating bench runs (box-to-box variation is ~1 %: compare inside one call)
#   usage (inside gpurun): SW="MMDA_FLAG_JOIN=0" BATCHES="32 256" bash tools/r3_ab.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SW=${SW:-MMDA_FLAG_JOIN=0}
for rep in 1 2; do
for cfg in "base:" "switch:$SW"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for B in ${BATCHES:-32 256}; do
    env $envs timeout -k 10 300 python bench.py --batch $B --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab_${name}_b${B}_$rep.json 2> gpurun_out/ab_${name}_b${B}_$rep.err
    echo "$name ($envs) B=$B rep=$rep $(python -c "import json,sys; d=json.load(open('gpurun_out/ab_${name}_b${B}_$rep.json')); print(d['ms_per_step'])")"
  done
done
done
