for s in 1 2 3 4; do
  echo "streams=$s: $(python bench.py --mode fwd --streams $s --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["dtype"], d["value"], d["ms_per_step"])')"
done
