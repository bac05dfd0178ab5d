import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma_valu.so"))
lib.mfma_valu_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
out = torch.empty(256 * 256, device=dev); clk = torch.zeros(256, dtype=torch.int64, device=dev)
src = torch.randn(512 * 8, device=dev).to(torch.bfloat16)
iters = 2000
def run(agpr, nf, ne):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = lib.mfma_valu_launch(agpr, nf, ne, 256, src.data_ptr(), out.data_ptr(), clk.data_ptr(), iters, st); assert rc == 0, rc
    torch.cuda.synchronize()
    return clk.double().median().item() / (iters * 4)
print("cycles per v_mfma_f32_32x32x16_bf16 (one wave per SIMD, all CUs busy, gaussian operands); NF v_fma + NE v_exp after every MFMA")
print("columns: acc in VGPRs | acc in ACC regs | acc in VGPRs, A and B from ACC regs (the QK form of attention_w64) | acc in ACC regs, A from ACC regs (its PV form)")
for ne in (0, 1, 2):
    for nf in (0, 2, 4, 5, 6, 8, 12):
        print("NE %d NF %2d: %6.1f %6.1f %6.1f %6.1f" % ((ne, nf) + tuple(run(m, nf, ne) for m in range(4))), flush=True)
