import sys, time, torch, os
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
data, _ = bench.build_batch(131072, 2048, 1000, dev)
csr = data.csr('edge_index2')
v = data.edge_attr2
for it in range(6):
    for k in list(csr._val_cache):
        if isinstance(k, tuple) and str(k[0]).startswith('y'): del csr._val_cache[k]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = csr.sym_index(v)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('sym_index total %.3f ms, unique share %.4f' % (dt * 1e3, s[0].numel() / csr.E))
