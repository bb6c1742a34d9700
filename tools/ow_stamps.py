import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd._lib import check, lib
from sparc_amd.ops import gram
dev = torch.device("cuda:0"); P = lambda t: ctypes.c_void_p(t.data_ptr())
m = 24000
for k in (16, 47, 64):
    Y = torch.randn(m, k, device=dev); G = gram(Y, Y)
    Wo = torch.empty(k, k, device=dev); Rc, Ri = (torch.empty(k, k, dtype=torch.float64, device=dev) for _ in range(2))
    sg = torch.empty(k, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        check(lib.sparc_orth_weights(m, k, P(G), P(Y), k, None, 0, P(Wo), P(Rc), P(Ri), P(sg), P(info), st), "fwd")
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 8)()
    lib.sparc_debug_orth_stamps(buf)
    t = list(buf)
    print(k, "phase cycles: load", t[1]-t[0], "chol", t[2]-t[1], "inverse", t[3]-t[2], "top+mm", t[4]-t[3], "lu", t[5]-t[4], "out", t[6]-t[5], "total", t[6]-t[0])
