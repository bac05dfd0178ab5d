# alternating A/B of an environment switch at the model level: tools/ab_env.sh "<ENV=VAL>" <mode> [rounds] [steps]
# prints ms/step of bench.py --mode <mode> with and without the switch, interleaved on the same box
sw=$1; mode=${2:-fwd}; n=${3:-3}; steps=${4:-20}
for i in $(seq $n); do
  a=$(python bench.py --mode $mode --steps $steps --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  b=$(env $sw python bench.py --mode $mode --steps $steps --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$mode base $a   $sw $b"
done
