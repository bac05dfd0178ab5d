for i in 1 2; do
for lib in "" "$PWD/aicity_action_amd/lib/pool_noslp.so"; do
  echo "lib=${lib##*/} train: $(MVIT_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-forward-record --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  echo "lib=${lib##*/} fwd: $(MVIT_HIP_LIB=$lib python bench.py --mode fwd --precision bf16 --no-cpu-baseline --steps 40 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
done
