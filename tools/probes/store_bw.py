import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "store_bw.so"))
lib.store_bw_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = ["row per lane, 32 rows x 32 B per instruction (GEMM epilogue)", "wave sub-tile linear, 5.33 rows x 192 B", "workgroup tile linear, 2.67 rows x 384 B",
         "pattern 0 into two separate matrices alternately", "pattern 0 into two matrices side by side in one [M][2N] buffer",
         "two separate matrices, all stores of the first then all of the second", "two separate matrices, alternating per 32 x 32 block"]
for M, N in ((802816, 384), (802816, 576), (200704, 768), (50176, 1536)):
    out = torch.empty(2 * M * N + 4096 * 64, dtype=torch.int16, device="cuda:0")
    for pat in range(7):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            assert lib.store_bw_launch(out.data_ptr(), M, N, pat, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            lib.store_bw_launch(out.data_ptr(), M, N, pat, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nb = M * N * 2 * (2 if pat >= 3 else 1)
        print("M=%d N=%d (%.0f MB) pattern %d (%s): %.1f us  %.2f TB/s" % (M, N, nb / 1e6, pat, names[pat], ms * 1e3, nb / ms / 1e9), flush=True)
