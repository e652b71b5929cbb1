"""CPU: host-side logic of the product package (no kernel launches): module surface == reference
surface, SpectralDesign against the golden vectors, collate/sharding, the C-ABI library loads and
exports what include/gml.h declares, and the GPU-only path refuses CPU tensors loudly."""
import ast
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import gnn_matlang_amd as G
from gnn_matlang_amd import _lib, models, synthetic
from gnn_matlang_amd.graph import collate, shard_graphs
from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = lambda a: torch.tensor(np.asarray(a))


def test_library_exports_match_header():
    hdr = open(os.path.join(ROOT, 'include', 'gml.h')).read()
    names = set(re.findall(r'\b(gml_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 15
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    L = ctypes.CDLL(_lib.LIB_PATH)                   # loads without a GPU
    for n in names:
        assert hasattr(L, n), n
    assert _lib.lib().gml_version() >= 1
    assert b'workspace' in _lib.lib().gml_error_string(-3)


def test_batch_descriptor_layout_matches_the_header(tmp_path):
    """gml_batch_desc (include/gml.h) is passed by pointer from ctypes (_lib.BatchDesc): same size, same field offsets, same flag
    values as the C compiler gives the header -- checked with gcc on the header itself (the header is plain C)."""
    import subprocess
    fields = [f[0] for f in _lib.BatchDesc._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gml.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(gml_batch_desc));\n' +
                   ''.join('  printf("%%zu\\n", offsetof(gml_batch_desc, %s));\n' % f for f in fields) +
                   '  printf("%d %d %d %d %d\\n", (int)GML_FWD_CHUNKED, (int)GML_DVAL_ACCUM, (int)GML_FWD_ONEWIN, (int)GML_POOL_SKIP_LAST, (int)GML_DMA_RING);\n'
                   '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split('\n')
    assert int(out[0]) == ctypes.sizeof(_lib.BatchDesc)
    for f, line in zip(fields, out[1:]):
        assert int(line) == getattr(_lib.BatchDesc, f).offset, f
    assert [int(v) for v in out[1 + len(fields)].split()] == [_lib.GML_FWD_CHUNKED, _lib.GML_DVAL_ACCUM, _lib.GML_FWD_ONEWIN,
                                                              _lib.GML_POOL_SKIP_LAST, _lib.GML_DMA_RING]


def test_cpu_tensors_fail_loudly():
    m = G.SpectConv(4, 3, 2, selfconn=False)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.randn(5, 4), torch.zeros(2, 0, dtype=torch.int64), torch.zeros(0, 2))
    l = G.ML3Layer(True, 2, 2, 4, 3, 2)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        l(torch.randn(5, 4), torch.zeros(2, 0, dtype=torch.int64), torch.zeros(0, 2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'gnn_matlang_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert '/root/reference' not in src.replace('/root/reference/', 'REF/'), f


def test_module_surface_matches_reference(golden):
    # ctor signature, attributes, repr, K handling (libs/spect_conv.py:26-56,101-103)
    m = G.SpectConv(7, 5, 3)
    assert m.selfconn and not m.depthwise and m.weight.shape == (4, 7, 5) and m.bias.shape == (5,)
    assert repr(m) == 'SpectConv(7, 5, K=4)'
    m = G.SpectConv(7, 5, 3, selfconn=False, depthwise=True, bias=False)
    assert m.weight.shape == (1, 7, 5) and m.DSweight.shape == (3, 7) and m.nsup == 3 and m.bias is None
    assert list(m.state_dict()) == ['DSweight', 'weight']
    assert float(m.DSweight.abs().sum()) == 0.0
    bound = (6.0 / (7 + 5)) ** 0.5
    assert float(m.weight.abs().max()) <= bound
    with pytest.raises(AssertionError):
        G.SpectConv(3, 3, 0)
    with pytest.raises(NotImplementedError):
        G.SpectConv(3, 3, 1, aggr='mean')
    c = G.SpectConCatConv(4, 6, 2)
    assert c.weight.shape == (3, 4, 6) and c.bias.shape == (18,)
    # ML3Layer state_dict keys/shapes == the reference's (captured in the golden fixture)
    g = golden('ml3layer.npz')
    for k in range(int(g['ncases'])):
        c = g.sub('case%03d/' % k)
        learnedge, ne, neo, ninp, nout1, nout2 = [int(v) for v in c['meta']]
        l = G.ML3Layer(bool(learnedge), ne, neo, ninp, nout1, nout2)
        ref = {n[len('param/'):]: v.shape for n, v in c.items() if n.startswith('param/')}
        assert {n: tuple(v.shape) for n, v in l.state_dict().items()} == ref
        assert l.learnedge == bool(learnedge) and l.nout2 == nout2
    # model assemblies load the reference state_dicts
    for f, ctor in (('model_zinc_gnnml3.npz', models.zinc_gnnml3), ('model_counting_gnnml3.npz', models.counting_gnnml3),
                    ('model_mutag_gnnml3.npz', models.mutag_gnnml3), ('model_mutag_gnnml1.npz', lambda: models.GNNML1Mutag(8))):
        ctor().load_state_dict({k: T(v) for k, v in golden(f).sub('param/').items()})
    assert sum(p.numel() for p in models.zinc_gnnml3().parameters()) == 33309
    assert sum(p.numel() for p in models.counting_gnnml3().parameters()) == 37649
    assert sum(p.numel() for p in models.sr25_gnnml3().parameters()) == 23714


def test_compat_import_path():
    from gnn_matlang_amd.libs.spect_conv import SpectConv, ML3Layer
    from gnn_matlang_amd.libs.utils import SpectralDesign, get_n_params
    assert SpectConv is G.SpectConv and ML3Layer is G.ML3Layer and SpectralDesign is G.SpectralDesign
    assert get_n_params(torch.nn.Linear(3, 2)) == 8


def test_spectral_design_matches_reference_vectors(golden):
    g = golden('spectral_design.npz')
    for k in range(int(g['ncases'])):
        c = g.sub('case%02d/' % k)
        kw = ast.literal_eval(str(c['kw']))
        d = G.SpectralDesign(**kw).design_many([(c['in_x'], c['in_edge_index'], 0)])[0]
        assert np.array_equal(d['edge_index2'], c['edge_index2']) and np.array_equal(d['x'], c['x'])
        np.testing.assert_allclose(d['edge_attr2'], c['edge_attr2'], rtol=0, atol=5e-6, err_msg=str(c['name']))
        np.testing.assert_allclose(d['lmax'], c['lmax'], rtol=1e-6)


def test_spectral_design_batched_equals_single_and_call_convention():
    raw = synthetic.make_graphs('zinc', 40, seed=5) + synthetic.make_graphs('counting', 10, seed=6)
    sd = G.SpectralDesign(recfield=2, dv=2, nfreq=7, adddegree=True)
    many = sd.design_many(raw)
    for (x, ei, y), d in zip(raw, many):
        one = sd.design_many([(x, ei, y)])[0]
        assert np.array_equal(one['edge_index2'], d['edge_index2'])
        np.testing.assert_allclose(one['edge_attr2'], d['edge_attr2'], rtol=0, atol=2e-6)

    class Data:
        pass
    dd = Data()
    dd.x, dd.edge_index = T(raw[0][0]), T(raw[0][1])
    out = sd(dd)
    assert out.edge_index2.dtype == torch.int64 and out.edge_attr2.dtype == torch.float32
    assert out.x.shape[1] == raw[0][0].shape[1] + 1 and out.edge_attr2.shape[1] == 8
    assert np.array_equal(out.edge_index2.numpy(), many[0]['edge_index2'])


def test_collate_and_shards():
    raw = synthetic.make_graphs('zinc', 9, seed=1)
    ds = G.SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(raw)
    b = collate(ds)
    n = [d['x'].shape[0] for d in ds]
    assert b.num_graphs == 9 and b.x.shape == (sum(n), 25) and b.ptr.tolist() == np.concatenate([[0], np.cumsum(n)]).tolist()
    assert b.edge_index2.shape[1] == b.edge_attr2.shape[0] and b.edge_attr2.shape[1] == 8
    # block diagonal: every edge stays inside its graph
    gid = b.batch
    assert bool((gid[b.edge_index2[0]] == gid[b.edge_index2[1]]).all())
    assert bool((gid[b.edge_index[0]] == gid[b.edge_index[1]]).all())
    cover = []
    for r in range(4):
        lo, hi = shard_graphs(9, r, 4)
        cover += list(range(lo, hi))
    assert cover == list(range(9))
    assert [shard_graphs(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_synthetic_generators_are_seeded_and_shaped():
    a = synthetic.make_graphs('zinc', 20, seed=0)
    b = synthetic.make_graphs('zinc', 20, seed=0)
    for (x1, e1, y1), (x2, e2, y2) in zip(a, b):
        assert np.array_equal(x1, x2) and np.array_equal(e1, e2) and y1 == y2
        n = x1.shape[0]
        assert 9 <= n <= 37 and x1.shape[1] == 25
        A = np.zeros((n, n)); A[e1[0], e1[1]] = 1
        assert np.array_equal(A, A.T) and A.sum(0).max() <= 4 and np.trace(A) == 0
    x, ei, y = synthetic.make_graphs('counting', 1, seed=2)[0]
    A = np.zeros((x.shape[0],) * 2); A[ei[0], ei[1]] = 1
    assert y == pytest.approx(np.trace(A @ A @ A) / 6)            # libs/utils.py:395-397
    x, ei, y = synthetic.make_graphs('mnist75', 1, seed=2)[0]
    assert x.shape == (75, 2)


# ------------------------------------------------------------------------------------------ batch plumbing (round 2)
def test_shard_graphs_balanced_tiles_and_balances():
    from gnn_matlang_amd.graph import shard_graphs_balanced
    rng = np.random.default_rng(0)
    for G, W in ((37, 1), (37, 2), (37, 8), (5, 8), (1000, 8), (0, 4)):
        w = rng.integers(1, 100, size=G)
        cuts = [shard_graphs_balanced(w, r, W) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == G
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(W - 1))         # tile [0, G): no gap, no overlap
        if G >= 100:                                                            # near-equal total work, unlike equal counts
            tot = np.array([w[a:b].sum() for a, b in cuts], dtype=np.float64)
            assert tot.max() - tot.min() <= 2 * w.max()
    assert shard_graphs_balanced(np.zeros(10), 1, 2) == (5, 10)                 # no work information: equal counts


def test_device_dataset_batches_equal_host_collate():
    """dataset.DeviceDataset (torch ops only: also runs on CPU tensors) assembles the same block-diagonal batch as
    graph.collate, for any selection and order of graphs."""
    import torch
    from gnn_matlang_amd import SpectralDesign, collate, synthetic
    from gnn_matlang_amd.dataset import DeviceDataset
    raw = synthetic.make_graphs('zinc', 12, seed=3)
    ds = SpectralDesign(recfield=2, dv=2, nfreq=3).design_many(raw)
    dd = DeviceDataset.from_graphs(ds, torch.device('cpu'))
    assert len(dd) == 12
    ids = [7, 0, 11, 3]
    got, ref = dd.batch(torch.tensor(ids)), collate([ds[i] for i in ids])
    for k in ('x', 'edge_index', 'edge_index2', 'edge_attr2', 'batch', 'ptr', 'y'):
        a, b = getattr(got, k), getattr(ref, k)
        assert a.shape == b.shape and torch.equal(a.to(b.dtype), b), k
    seen = torch.cat([b.y for b in dd.epoch(5, generator=torch.Generator().manual_seed(0))])
    assert seen.numel() == 12 and torch.equal(torch.sort(seen)[0], torch.sort(dd.y)[0])   # one epoch = every graph once


def test_raw_readers_reproduce_the_reference_datasets():
    """readers.py (numpy only) on the raw files the reference ships (tests/golden/raw: mutag.mat 22 KB, sr251256.g6 780 B)
    against the arrays the golden generator extracted with scipy / networkx through the reference's own ``process``
    bodies (libs/utils.py:192-209, 506-513)."""
    from gnn_matlang_amd import readers
    raw = os.path.join(GOLDEN, 'raw')
    for fname, loader, gold in (('mutag.mat', readers.load_mutag, 'data_mutag.npz'),
                                ('sr251256.g6', readers.load_sr, 'data_sr25.npz')):
        graphs = loader(os.path.join(raw, fname))
        z = np.load(os.path.join(GOLDEN, gold))
        nptr, eptr = z['node_ptr'], z['edge_ptr']
        assert len(graphs) == len(nptr) - 1
        for i, (x, ei, y) in enumerate(graphs):
            assert np.array_equal(x, z['x'][nptr[i]:nptr[i + 1]]) and x.dtype == np.float32, (fname, i)
            assert np.array_equal(ei, z['edge_index'][:, eptr[i]:eptr[i + 1]]) and ei.dtype == np.int64, (fname, i)
            assert float(y) == float(z['y'][i])


def test_dense_block_row_slabs():
    """dense_block._splits: the weight-gradient GEMM is cut into <= 64 row slabs of >= 1024 rows that divide the row count."""
    from gnn_matlang_amd.dense_block import _splits
    for rows in (76800, 307200, 75 * 125, 75, 1024, 2048, 75 * 1000, 97 * 75):
        p = _splits(rows)
        assert 1 <= p <= 64 and rows % p == 0
        assert p == 1 or rows // p >= 1024
    assert _splits(76800) == 64 and _splits(75) == 1


def test_chain_after_bookkeeping():
    """ML3Layer.chain_after (host logic of the relu hand-over and the stacked edge branch): the hand-over token of the layer
    below is offered only for the very tensor object that layer returned, under grad mode, to a tensor that requires grad; the
    declaration does not register submodules, survives deepcopy, can be withdrawn; the stash key of the stacked edge branch
    follows the weights' versions."""
    import copy
    import weakref
    from gnn_matlang_amd import models, functional as Fn
    from gnn_matlang_amd import spect_conv as SC
    net = models.GNNML3(32, 8, 30, 2, 3)
    c1, c2, c3 = net.conv1, net.conv2, net.conv3
    assert c2._chain_prev[0] is c1 and c3._chain_prev[0] is c2 and c1._chain_next[0] is c2 and not c1._chain_prev
    assert len([n for n, _ in net.named_modules() if n.startswith('conv2.')]) == len([n for n, _ in net.named_modules() if n.startswith('conv1.')])
    twin = copy.deepcopy(net)
    assert twin.conv2._chain_prev[0] is twin.conv1 and twin.conv2._chain_prev[0] is not c1
    assert not models.GNNML3(32, 8, 30, 2, 3, bn=True).conv2._chain_prev       # BatchNorm in between: another tensor, not declared
    # the token is offered for the declared layer's own output tensor only
    out = torch.zeros(5, 32, requires_grad=True)
    tok = Fn.ChainToken(30)
    SC._CHAIN_STATE[c1] = (weakref.ref(out), tok)
    cin, cout = c2._chain_args(out)
    assert cin is tok and cout is not tok and cout.cols == 30 and not cout.premasked
    assert c2._chain_args(out.clone())[0] is None                          # a different tensor object
    assert c2._chain_args(out.detach())[0] is None                         # no gradient wanted
    with torch.no_grad():
        assert c2._chain_args(out)[0] is None
    assert c1._chain_args(out)[0] is None                                  # nothing declared below conv1
    c2.chain_after(None)
    assert c2._chain_args(out)[0] is None and not c1._chain_next
    c2.chain_after(c1)
    # stash key of the stacked edge branch: supports' identity + the four weights' versions
    val = torch.zeros(7, 8)
    k0 = c2._edge_key(val, 'csr')
    assert k0 == c2._edge_key(val, 'csr') and k0 != c2._edge_key(val.clone(), 'csr') and k0 != c2._edge_key(val, 'other')
    with torch.no_grad():
        c2.fc1_3.weight.add_(1.0)
    assert c2._edge_key(val, 'csr') != k0


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location('gml_bench', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_line_is_compact():
    """a 24 KB line made the driver's record unparseable.  The contract line is built from the full record by
    bench.compact_line: json.loads round-trips, < 8,000 characters (budget 6,000), roofline + cpu_baseline + config kept
    whatever else the record grows."""
    import json
    bench = _load_bench()
    long = 'x' * 3000
    res = {'metric': 'GNNML3 training graphs/sec on ZINC-12k', 'value': 14182965.75530812, 'unit': 'graphs/s', 'n_gpus': 1,
           'steps': 20, 'warmup': 5, 'ms_per_step': 9.241508599916415, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': long, 'graphs_per_gpu': 131072, 'parallelism': 'dp1'},
           'roofline': {'bound': 'hbm', 'achieved': 2518.5193540469654, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.3148149192558707,
                        'traffic': 2560695226.6666665, 'kernel': long, 'traffic_source': long, 'note': long},
           'cpu_baseline': {'value': 6320.123172658106, 'unit': 'graphs/s', 'cores': 16, 'kind': 'port', 'sample': long,
                            'thread_ladder': [{'threads': t, 'value': 1.0} for t in range(200)]},
           'other_configs': [{'note': long}] * 10, 'sr25_sweep': [{'note': long}] * 10, 'mnist75': {'note': long},
           'parity_vs_oracle': {'note': long}, 'parity_vs_oracle_after_training': {'note': long},
           'block_seconds': [0.1] * 500, 'per_rank_ms_per_step': [9.0] * 8,
           'max_rel_err_vs_oracle': {m: {'gradients_termsum': 1e-5, 'logits': 1e-6} for m in ('a', 'b', 'c')},
           'spmm': {'bound': 'hbm', 'achieved': 6025.0, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.753, 'block_ms': [0.7] * 100},
           'epoch_bs64': {'value': 244599.0, 'unit': 'graphs/s', 'ms_per_step': 0.26, 'mode': long, 'eager': {'note': long}},
           'value_exact_fp32': {'value': 5.6e6, 'unit': 'graphs/s', 'ms_per_step': 23.3, 'arithmetic': long}}
    line = bench.compact_line(res)
    assert '\n' not in line and len(line) <= bench.LINE_BUDGET < 8000
    back = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in back, k
    assert abs(back['value'] - res['value']) <= 1e-5 * res['value']
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(back['roofline'])
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(back['cpu_baseline'])
    assert 'other_configs' not in back and 'sr25_sweep' not in back and 'block_seconds' not in back
    # a record stuffed far beyond the budget still yields a parseable line with the mandatory objects
    res['max_rel_err_vs_oracle_after_training'] = {('mode%d' % i): {'gradients_termsum': 1e-5, 'logits': 1e-6} for i in range(200)}
    line = bench.compact_line(res)
    back = json.loads(line)
    assert len(line) <= bench.LINE_BUDGET and 'roofline' in back and 'cpu_baseline' in back and 'config' in back
    # and the real round-5 record (24 KB) comes out well inside the budget
    rec = os.path.join(ROOT, 'profiles', 'r05_f_bench_default.json')
    if os.path.exists(rec):
        line = bench.compact_line(json.load(open(rec)))
        assert len(line) <= bench.LINE_BUDGET and json.loads(line)['roofline']['frac'] > 0
