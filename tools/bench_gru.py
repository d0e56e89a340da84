"""GRU recurrence kernels alone (HIP events): python tools/bench_gru.py [other libsept .so to compare]"""
import ctypes, os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
libs = [os.path.join(ROOT, "speech-emotion-privacy-trust_amd/csrc/libsept_hip.so")] + sys.argv[1:]
B, T, H = 224, 25, 64
g = torch.Generator().manual_seed(1)
gi = torch.randn(B, T, 2, 3 * H, generator=g).cuda()
whh = [(torch.randn(3 * H, H, generator=g) * 0.1).cuda() for _ in range(2)]
bhh = [(torch.randn(3 * H, generator=g) * 0.1).cuda() for _ in range(2)]
out = torch.empty(B, T, 2 * H, device="cuda"); gates = torch.empty(B, T, 2, 4, H, device="cuda")
dout = torch.randn(B, T, 2 * H, generator=g).cuda()
dgi = torch.empty(B, T, 2, 3 * H, device="cuda"); dgh = torch.empty_like(dgi); hprev = torch.empty(B, T, 2, H, device="cuda")
P = ctypes.c_void_p
res = {}
for path in libs:
    lib = ctypes.CDLL(path)
    lib.sept_gru_forward.argtypes = [P] * 7 + [ctypes.c_int] * 3 + [P]
    lib.sept_gru_backward.argtypes = [P] * 8 + [ctypes.c_int] * 3 + [P]
    st = torch.cuda.current_stream().cuda_stream
    fwd = lambda: lib.sept_gru_forward(gi.data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), bhh[0].data_ptr(), bhh[1].data_ptr(), out.data_ptr(), gates.data_ptr(), B, T, H, st)
    bwd = lambda: lib.sept_gru_backward(dout.data_ptr(), out.data_ptr(), gates.data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), dgi.data_ptr(), dgh.data_ptr(), hprev.data_ptr(), B, T, H, st)
    for name, fn in (("forward", fwd), ("backward", bwd)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        print(f"{os.path.basename(os.path.dirname(path))}/{os.path.basename(path)} gru {name}: {s.elapsed_time(e) / 50 * 1e3:.1f} us")
    res[path] = (out.clone(), dgi.clone())
if len(libs) == 2:
    a, b = res[libs[0]], res[libs[1]]
    print("max |d out|", float((a[0] - b[0]).abs().max()), "max |d dgi|", float((a[1] - b[1]).abs().max()))
