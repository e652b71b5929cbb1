import sys, time, torch
sys.path.insert(0, '/root/repo')
from gnn_matlang_amd import functional as Fn
dev = torch.device('cuda:0')
for S in (24, 48):
    E = 2_000_000
    torch.manual_seed(0)
    ea = torch.randn(E, S, device=dev) * 0.7
    ws = [(torch.randn(2 * S, S, device=dev) / S ** 0.5).requires_grad_(True) for _ in range(3)] + [(torch.randn(S, 4 * S, device=dev) / (4 * S) ** 0.5).requires_grad_(True)]
    go = torch.randn(E, S, device=dev)
    for lib in (False, True):
        Fn.EDGE_WIDE_BWD_LIB = lib
        def step():
            for w in ws: w.grad = None
            out = Fn.EdgeBranchWide.apply(ea, *ws)
            out.backward(go)
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print('S=%d E=%d %s: fwd+bwd %.2f ms' % (S, E, 'library backward' if lib else 'HIP backward', dt * 1e3), flush=True)
