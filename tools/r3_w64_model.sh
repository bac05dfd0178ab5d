for w in 0 1 0 1; do
  echo "MVIT_ATT_W64=$w fwd fp16: $(MVIT_ATT_W64=$w python bench.py --mode fwd --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"])')"
  echo "MVIT_ATT_W64=$w fwd bf16: $(MVIT_ATT_W64=$w python bench.py --mode fwd --precision bf16 --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"])')"
  echo "MVIT_ATT_W64=$w train: $(MVIT_ATT_W64=$w python bench.py --no-cpu-baseline --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
MVIT_ATT_W64=1 python -m pytest tests/test_hip_model.py tests/test_hip_ops.py tests/test_hip_train.py -q -x 2>&1 | tail -3
