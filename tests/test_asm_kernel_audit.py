"""CPU: the kernels that own ACC registers by name in their asm text (csrc/attention_w64.hip, mlp_fused.hip) are compiled to
assembly for both 16-bit builds and audited the way cdna_hip_programming.md section 5.7 item 4 asks after every edit: the compiler must
not touch an ACC register itself (a spill into a[0:239] is silent corruption of the accumulators), must not spill, must not use scratch.
"""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aicity_action_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-fno-honor-nans", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def _audit(args):
    src, define = args
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [HIPCC] + FLAGS + ([define] if define else []) + [os.path.join(CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
    inasm, bad = False, []
    for line in text.split("\n"):
        if "ASMSTART" in line:
            inasm = True
        elif "ASMEND" in line:
            inasm = False
        elif not inasm and ("v_accvgpr" in line or "scratch_" in line):
            bad.append(line.strip())
    spills = [int(m) for m in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
    scratch = [int(m) for m in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
    kernels = re.findall(r"\.agpr_count:\s+(\d+)", text)
    return src, define, bad, spills, scratch, kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_asm_owned_acc_registers_are_left_alone_by_the_compiler():
    jobs = [(s, d) for s in ("attention_w64.hip", "mlp_fused.hip") for d in ("", "-DMVIT_HALF_IS_FP16")]
    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(_audit, jobs))
    for src, define, bad, spills, scratch, kernels in results:
        tag = "%s %s" % (src, define or "(bf16)")
        own = 256 if src == "mlp_fused.hip" else 240          # (mlp_fused.hip also holds its small packing kernels: 0 ACC registers)
        assert kernels and all(int(k) in (own, 0) for k in kernels) and any(int(k) == own for k in kernels), \
            "%s: expected kernels that own a[0:%d], got agpr counts %s" % (tag, own - 1, kernels)
        assert not bad, "%s: compiler-generated ACC / scratch instructions outside the asm statements: %s" % (tag, bad[:5])
        assert spills and max(spills) == 0, "%s: VGPR spills %s" % (tag, spills)
        assert scratch and max(scratch) == 0, "%s: scratch memory %s" % (tag, scratch)
