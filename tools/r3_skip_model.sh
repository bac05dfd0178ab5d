python -m pytest tests/test_hip_model.py tests/test_hip_train.py tests/test_hip_engine.py -q -x 2>&1 | tail -3
for f in 0 1 0 1; do
  echo "MVIT_SKIP_FUSE=$f train: $(MVIT_SKIP_FUSE=$f python bench.py --no-cpu-baseline --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  echo "MVIT_SKIP_FUSE=$f fwd bf16: $(MVIT_SKIP_FUSE=$f python bench.py --mode fwd --precision bf16 --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
MVIT_SKIP_FUSE=1 bash tools/prof_noside.sh r3_skip1 > gpurun_out/r3_skip1.txt 2>&1
python3 tools/kstats.py gpurun_out/r3_skip1 7 80 | grep -iE "maxpool|linear_mfma|wgrad_mfma|proj_max|sum of" | cut -c1-200
