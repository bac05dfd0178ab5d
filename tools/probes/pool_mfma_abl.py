import ctypes, os, sys, torch
here = "tools/probes"
dev = "cuda:0"
B, h, T, H, W = 8, 4, 8, 28, 28
C = 96 * h; N = T * H * W
qkv = torch.randn(B, N, 3 * C, device=dev).to(torch.bfloat16); w = torch.randn(96, 27, device=dev) * 0.3
out = torch.zeros(B * h, N, 96, device=dev, dtype=torch.bfloat16)
st = lambda: torch.cuda.current_stream().cuda_stream
for name in ("pool_mfma", "pool_mfma_NOSTORE", "pool_mfma_NOSTAGE", "pool_mfma_NOMFMA"):
    lib = ctypes.CDLL(os.path.join(here, name + ".so"))
    lib.pool_mfma_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    fn = lambda: lib.pool_mfma_probe_launch(qkv.data_ptr(), 3 * C, 0, h, w.data_ptr(), out.data_ptr(), B * h, st())
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); e1.synchronize()
    print("%-20s %.1f us" % (name, e0.elapsed_time(e1) / 50 * 1e3))
