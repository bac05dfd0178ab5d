# GPU box: inference sub-batch streams (HIP.STREAMS) 1..4, forward fp16 / bf16 and the sliding window, one session
for s in 2 3 4 2 3; do
  echo "streams=$s fwd fp16: $(python bench.py --mode fwd --streams $s --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  echo "streams=$s fwd bf16: $(python bench.py --mode fwd --precision bf16 --streams $s --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
for s in 2 3; do
  echo "streams=$s window: $(python bench.py --mode window --streams $s --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
