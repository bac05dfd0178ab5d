# forward clips/s for different numbers of sub-batch streams (HIP.STREAMS)
for s in "$@"; do
  python bench.py --mode fwd --no-cpu-baseline --no-kernel-timing --streams $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s', d['value'], d['ms_per_step'])"
done
