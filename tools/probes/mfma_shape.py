import ctypes, os, time, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma_shape.so"))
lib.mfma_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
out = torch.empty(512 * 512, device=dev)
clk = torch.zeros(512 * 2, dtype=torch.int64, device=dev)
gauss = torch.randn(4096 * 8, device=dev).to(torch.bfloat16)
zero = torch.zeros(4096 * 8, device=dev, dtype=torch.bfloat16)
iters = 4000
flops_wave = iters * 8 * 4 * 32768        # per wave: 64x64 tile, 8 k-steps of 16 per iteration
def run(shape, lds, threads, blocks, data, secs=2.0):
    st = torch.cuda.current_stream().cuda_stream
    def go(n):
        for _ in range(n):
            rc = lib.mfma_probe_launch(shape, lds, threads, blocks, data.data_ptr(), out.data_ptr(), clk.data_ptr(), iters, st)
            assert rc == 0, rc
    go(3); torch.cuda.synchronize()
    t0 = time.time(); go(10); torch.cuda.synchronize(); per = (time.time() - t0) / 10
    n = max(10, int(secs / per))
    go(n); torch.cuda.synchronize()           # sustained load first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(20); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    c = clk.view(-1, 2)[:blocks].double()
    cyc, rt = c[:, 0].median().item(), c[:, 1].median().item()
    tf = flops_wave * (threads // 64) * blocks / (ms * 1e-3) / 1e12
    return ms, tf, cyc, cyc / rt * 100.0
for data, dn in ((gauss, "gaussian"), (zero, "zero")):
    for lds in (0, 1):
        for threads, blocks, wn in ((256, 256, "1 wave/SIMD"), (512, 256, "2 waves/SIMD")):
            res = []
            for shape in (0, 1):
                ms, tf, cyc, mhz = run(shape, lds, threads, blocks, data)
                nm = iters * 32 * (1 if shape == 0 else 2) * (threads // 256)
                res.append(tf)
                print("%-8s %-9s %-13s %-9s: %7.3f ms  %7.1f TFLOP/s  clock %6.0f MHz  %5.1f cycles per MFMA per SIMD" % (
                    dn, "LDS reads" if lds else "registers", wn, "32x32x16" if shape == 0 else "16x16x32", ms, tf, mhz, cyc / nm), flush=True)
            print("    ratio 16x16x32 / 32x32x16 = %.3f" % (res[1] / res[0]), flush=True)
