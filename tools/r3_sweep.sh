# round-3 configuration sweep (same flags as DESIGN section 4's list): ms/step of every shape / mode the bench can run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err; python - <<PY
import json
try:
    j=json.load(open("gpurun_out/sw_$name.json")); print("$name", j["ms_per_step"], "ms", round(j["value"]), "samples/s", j["roofline"]["all_launch_ms"])
except Exception as e:
    print("$name FAILED", e)
PY
}
run b32
run b64 --batch 64
run b128 --batch 128
run b256 --batch 256
run t500 --seq-len 500 --steps 50 --warmup 5
MMDA_GEMM_DMA=0 run t500_nodma --seq-len 500 --steps 50 --warmup 5
run ragged_confid --ragged 1 --confidnet 1
run gru --rnncell gru
run c5 --confidnet 1 --fp8-fusion 1
run c5_b256 --confidnet 1 --fp8-fusion 1 --batch 256
run fp32 --precision fp32 --steps 50 --warmup 5
