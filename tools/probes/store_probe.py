import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
hip = ctypes.CDLL("libamdhip64.so")
mod = ctypes.c_void_p(); fn = ctypes.c_void_p()
# load the kernel through the HIP module API from the fat binary is awkward; simpler: the .so registers the kernel, launch via hipLaunchKernel on its symbol
lib = ctypes.CDLL(os.path.join(here, "store_probe.so"))
out = torch.empty(256 * 8 * (4 << 20), dtype=torch.uint8, device="cuda:0")
cyc = torch.zeros(256 * 8, device="cuda:0")
sym = ctypes.cast(lib.store_probe, ctypes.c_void_p)
class dim3(ctypes.Structure):
    _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]
hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, dim3, dim3, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
def run(pattern, rounds, stride, nw):
    a = [ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), ctypes.c_int(pattern), ctypes.c_int(rounds), ctypes.c_int(stride), ctypes.c_int(nw)]
    arr = (ctypes.c_void_p * len(a))(*[ctypes.cast(ctypes.pointer(x), ctypes.c_void_p) for x in a])
    cyc.zero_()
    for _ in range(2):
        rc = hip.hipLaunchKernel(sym, dim3(256, 1, 1), dim3(512, 1, 1), arr, 0, None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    c = cyc.view(256, 8)[:, :nw]
    return c.mean().item()
names = ["16 rows x 64 B", "1 KiB contiguous", "8 rows x 128 B", "64 rows x 16 B", "16 rows x 64 B (row-adjacent lanes)"]
for nw in (8, 4, 1):
    for stride in (768, 3072):
        for p in range(5):
            c = run(p, 64, stride, nw)
            print("waves %d stride %5d  %-36s: %7.0f cycles per 12 stores per wave -> %.1f cycles per store instruction per CU, %.1f B/clk/CU" % (
                nw, stride, names[p], c, c / 12 / nw, 12 * nw * 1024 / c))
