import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "valu_rate.so"))
lib.valu_rate_launch.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
out = torch.empty(256 * 1024, device=dev); clk = torch.zeros(256, dtype=torch.int64, device=dev)
src = torch.randn(256, device=dev).to(torch.bfloat16).view(torch.int16)
iters = 2000
names = ["v_fma_f32", "v_pk_fma_f32", "v_dot2_f32_bf16", "v_dot2_f32_f16", "v_dot2c_f32_bf16", "v_perm_b32", "v_fmac_f32", "v_lshlrev_b32", "v_dot2c_f32_f16"]
print("cycles (s_memtime, 100 MHz ticks x clock ratio folded out against v_fma_f32) per instruction per wave; waves per SIMD 1 / 2 / 4")
base = None
for op, n in enumerate(names):
    row = []; wall = []
    for w in (1, 2, 4):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            rc = lib.valu_rate_launch(op, w, src.data_ptr(), out.data_ptr(), clk.data_ptr(), iters, st); assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.valu_rate_launch(op, w, src.data_ptr(), out.data_ptr(), clk.data_ptr(), iters, st)
        e1.record(); torch.cuda.synchronize()
        row.append(clk.double().median().item() / (iters * 32))
        wall.append(e0.elapsed_time(e1) * 1e6 / (iters * 32))      # ns per instruction per wave (launch overhead included)
    if base is None: base = row
    print("%-18s ticks/instr %s   relative to v_fma_f32 %s" % (n, " ".join("%7.3f" % r for r in row), " ".join("%5.2f" % (r / b) for r, b in zip(row, base))) + "   wall ns/instr " + " ".join("%6.3f" % x for x in wall), flush=True)
