"""gnn_matlang_amd -- the GNNML1/GNNML3 spectral message-passing layer of balcilar/gnn-matlang
(libs/spect_conv.py: SpectConv / ML3Layer) on AMD MI355X (gfx950).

    from gnn_matlang_amd import SpectConv, ML3Layer, SpectralDesign       # reference API
    from gnn_matlang_amd.libs.spect_conv import SpectConv, ML3Layer       # reference import path

Host code is Python on PyTorch-ROCm; the arithmetic is hand-written HIP in libgml_hip.so (C ABI in
include/gml.h), loaded lazily with ctypes on the first forward.  No CPU fallback exists.
"""
from .spect_conv import SpectConv, SpectConCatConv, ML3Layer, glorot, zeros
from .spectral_design import SpectralDesign
from .graph import GraphCSR, Batch, collate, csr_for, shard_graphs, shard_graphs_balanced
from .dataset import DeviceDataset

__all__ = ['SpectConv', 'SpectConCatConv', 'ML3Layer', 'SpectralDesign', 'GraphCSR', 'Batch', 'collate',
           'csr_for', 'shard_graphs', 'shard_graphs_balanced', 'DeviceDataset', 'glorot', 'zeros']
