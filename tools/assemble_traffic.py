"""Builds profiles/r5_{attn_fwd,attn_bwd,block_tail}_hbm_traffic.json (the files bench.py's `roofline.traffic` reads) from the raw outputs of
tools/r5_profiles.sh in gpurun_out/:   python tools/assemble_traffic.py <prefix> <commit> [out-prefix = r5]"""
import json, os, re, sys
pre, commit = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else "r5"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda n: os.path.join(root, "gpurun_out", n)


def load(n):
    return json.load(open(g(n)))


fwd = {"by_clips_per_launch": {}, "measured_at_commit": commit}
for clips, name in ((3, "%s_attn_fwd3_hbm_traffic.json" % pre), (4, "%s_attn_fwd4_hbm_traffic.json" % pre), (8, "%s_attn_fwd8_hbm_traffic.json" % pre)):
    if not os.path.exists(g(name)):
        continue
    d = load(name)
    d["workload"] = ("bench.py --mode fwd --precision bf16, @448; attn_fwd_w64_kernel launches (one per mvit_attention_fwd call); 4 = a sub-batch of the default "
                     "HIP.STREAMS 2 at B=8, 3 = HIP.STREAMS 3 (measured with --batch 9: every launch 3 clips), 8 = --streams 1 at B=8")
    d["clips_per_launch"] = clips
    fwd["by_clips_per_launch"][str(clips)] = d
json.dump(fwd, open(os.path.join(root, "profiles", "%s_attn_fwd_hbm_traffic.json" % out), "w"), indent=1)
d = load("%s_attn_bwd_hbm_traffic.json" % pre)
d["workload"] = ("bench.py (train), B=8 @448 bf16; kernels attn_bwd_delta (which also writes the pre-scaled 16-bit queries) + attn_bwd_dq + attn_bwd_dkv "
                 "(+ dkv slab reduce on the two split launches) of one mvit_attention_bwd call")
d["clips_per_launch"] = 8
json.dump({"by_clips_per_launch": {"8": d}, "measured_at_commit": commit}, open(os.path.join(root, "profiles", "%s_attn_bwd_hbm_traffic.json" % out), "w"), indent=1)
tail = {"by_clips_per_launch": {}, "measured_at_commit": commit}
for clips, shp in ((8, "50176x384"), (4, "25088x384"), (3, "18816x384")):
    if not os.path.exists(g("%s_pmc_block_tail_%s.txt" % (pre, shp))):
        continue
    txt = open(g("%s_pmc_block_tail_%s.txt" % (pre, shp))).read()
    m = re.search(r"fetch ([0-9.]+) MB.*write ([0-9.]+) MB; algorithmic ([0-9.]+) MB", txt)
    f, w, a = (float(x) * 1e6 for x in m.groups())
    tail["by_clips_per_launch"][str(clips)] = {
        "kernel": "mlp_fused_kernel<12, 1, true> (mvit_block_tail_fwd)", "shape": shp, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
        "traffic_bytes_per_launch": f + w, "algorithmic_bytes_per_launch": a, "clips_per_launch": clips,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/block_tail_bench.py; FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1"}
json.dump(tail, open(os.path.join(root, "profiles", "%s_block_tail_hbm_traffic.json" % out), "w"), indent=1)
print("wrote profiles/%s_{attn_fwd,attn_bwd,block_tail}_hbm_traffic.json" % out)
