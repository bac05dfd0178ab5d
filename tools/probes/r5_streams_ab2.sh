rm -rf gpurun_out/traffic
MARKER=attn_bwd_dq_kernel tools/traffic.sh attn_bwd gpurun_out/r5_final2_attn_bwd_hbm_traffic.json --mode train > /dev/null 2>&1
rm -rf gpurun_out/traffic
cat gpurun_out/r5_final2_attn_bwd_hbm_traffic.json | head -14
for rep in 1 2 3 4 5; do for st in 2 3; do
  echo "HIP.STREAMS $st fwd fp16: $(python bench.py --mode fwd --streams $st --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
