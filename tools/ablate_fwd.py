#!/usr/bin/env python3
"""Micro-benchmark of the fused forward kernel alone (C ABI), with the debug ablation bits."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_batch
from gnn_matlang_amd import functional as Fn

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
data, _ = build_batch(B, 2048, 1000, dev)
csr = data.csr()
N, E = csr.N, csr.E
val = csr.sort_values(data.edge_attr2)
x = torch.randn(N, 32, device=dev)
w = torch.randn(8, 32, 30, device=dev) * 0.1
b = torch.randn(30, device=dev)
out = torch.empty(N, 32, device=dev)
q, f = Fn.conv_cost(N, E, 8, 32, 30)
print('N %d E %d  Q %.1f MB  F %.2f GFLOP' % (N, E, q / 1e6, f / 1e9))
for name, flags in (('full', 1), ('no_mfma', 1 | 0x100), ('no_agg', 1 | 0x200), ('no_stage', 1 | 0x400),
                    ('no_agg_no_mfma', 1 | 0x300), ('only_barriers', 1 | 0x700), ('+no_epilogue', 1 | 0xf00), ('+no_colstage', 1 | 0x1f00), ('+no_wstage', 1 | 0x3f00), ('full_no_epi', 1 | 0x800)):
    for it in range(3):
        Fn._fused_conv(csr.rowptr, csr.col, csr.ginfo, None, val, x, 32, w, (960, 30, 1), b, out, 32, N, 8, 32, 30, flags, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(10):
        Fn._fused_conv(csr.rowptr, csr.col, csr.ginfo, None, val, x, 32, w, (960, 30, 1), b, out, 32, N, 8, 32, 30, flags, 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('%-16s %.3f ms   %.1f TFLOP/s  %.0f GB/s' % (name, ms, f / ms / 1e9, q / ms / 1e6))
