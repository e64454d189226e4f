#!/bin/bash
# Diagnostics: time bench.py with each shared library under build_variants/ in place of mmda_amd/libmmda_hip.so (on the GPU box's copy
# of the tree only).  usage (inside gpurun): bash tools/bench_variants.sh [batch]
B=${1:-32}
cp mmda_amd/libmmda_hip.so /tmp/base.so
for v in /tmp/base.so build_variants/*.so; do
  cp $v mmda_amd/libmmda_hip.so
  python bench.py --batch $B --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['ms_per_step'], list(d['roofline']['all_launch_ms'].values()))" || exit 1
done
cp /tmp/base.so mmda_amd/libmmda_hip.so
