cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "c3_ or c5_as_one" > gpurun_out/t_pair.log 2>&1; echo "rc=$?" >> gpurun_out/t_pair.log
tail -5 gpurun_out/t_pair.log
for i in 1 2; do
echo -n "pair: "; timeout -k 10 300 python bench.py --no-cpu-baseline --batch 256 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['all_launch_ms'])"
echo -n "nopair: "; MMDA_LSTM_PAIR=0 timeout -k 10 300 python bench.py --no-cpu-baseline --batch 256 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['all_launch_ms'])"
done
