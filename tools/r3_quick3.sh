cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "scatter or embedding or one_rank or dp_step or golden or train_epoch or shape_changes or degenerate" > gpurun_out/t_sel.log 2>&1; echo "rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - <<PY
import json
try:
    j=json.load(open("gpurun_out/sw_$name.json")); print("$name", j["ms_per_step"], "ms", round(j["value"]), "samples/s")
except Exception as e:
    print("$name FAILED", e)
PY
}
run b32
run ragged --ragged 1
run ragged_confid --ragged 1 --confidnet 1
run confid --confidnet 1
run ragged_b256 --ragged 1 --batch 256
