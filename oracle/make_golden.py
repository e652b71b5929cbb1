#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Runs only in the build container
(it needs /root/reference, which does not exist on the GPU box); its OUTPUTS
(.npz: inputs + expected outputs) are committed, the reference is not.

How the reference is run: /root/reference/libs/spect_conv.py and libs/utils.py are
imported UNMODIFIED.  Their third-party dependency torch_geometric (==1.6.1 per
/root/reference/README.md:9-11) is not installed and cannot be (no network), so
a minimal stand-in module is put in ``sys.modules`` that provides exactly the
slice the two files touch: ``MessagePassing`` whose ``propagate`` is PyG 1.6.1's
documented gather -> message -> scatter-add for aggr='add',
flow='source_to_target'; ``OptTensor``; inert ``utils`` helpers; ``Data`` /
``InMemoryDataset`` shells.  (The reference has no tests or golden vectors of its
own -- SURVEY s4 -- so these generated vectors are what pins the oracle.)

Usage:  python oracle/make_golden.py      (rewrites tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


# ------------------------------------------------------------------ stand-in PyG
def install_pyg_standin():
    tg = types.ModuleType('torch_geometric')
    typing_m = types.ModuleType('torch_geometric.typing')
    typing_m.OptTensor = 'Optional[torch.Tensor]'
    nn_m = types.ModuleType('torch_geometric.nn')
    conv_m = types.ModuleType('torch_geometric.nn.conv')
    utils_m = types.ModuleType('torch_geometric.utils')
    data_m = types.ModuleType('torch_geometric.data')
    data_data_m = types.ModuleType('torch_geometric.data.data')

    class MessagePassing(torch.nn.Module):
        def __init__(self, aggr='add', flow='source_to_target', node_dim=0):
            super().__init__()
            assert aggr == 'add' and flow == 'source_to_target' and node_dim == 0
            self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

        def propagate(self, edge_index, size=None, **kwargs):
            x = kwargs['x']
            x_j = x.index_select(0, edge_index[0])
            msg = self.message(x_j, kwargs['norm'])
            out = torch.zeros(x.size(0), msg.size(1), dtype=msg.dtype)
            return out.scatter_add_(0, edge_index[1].view(-1, 1).expand_as(msg), msg)

    class Data(object):
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class InMemoryDataset(object):
        pass

    def _inert(*a, **k):
        raise NotImplementedError('stand-in')

    conv_m.MessagePassing = MessagePassing
    for name in ('remove_self_loops', 'add_self_loops', 'get_laplacian', 'to_networkx', 'to_undirected'):
        setattr(utils_m, name, _inert)
    data_m.InMemoryDataset, data_m.Data = InMemoryDataset, Data
    data_data_m.Data = Data
    tg.typing, tg.nn, tg.utils, tg.data = typing_m, nn_m, utils_m, data_m
    nn_m.conv = conv_m
    data_m.data = data_data_m
    for k, m in {'torch_geometric': tg, 'torch_geometric.typing': typing_m, 'torch_geometric.nn': nn_m,
                 'torch_geometric.nn.conv': conv_m, 'torch_geometric.utils': utils_m,
                 'torch_geometric.data': data_m, 'torch_geometric.data.data': data_data_m}.items():
        sys.modules[k] = m
    return Data


Data = install_pyg_standin()
sys.path.insert(0, REF)
from libs.spect_conv import SpectConv as RefSpectConv, ML3Layer as RefML3Layer, \
    SpectConCatConv as RefSpectConCatConv          # noqa: E402
from libs.utils import SpectralDesign as RefSpectralDesign   # noqa: E402
sys.path.remove(REF)
for _k in [k for k in sys.modules if k == 'libs' or k.startswith('libs.')]:
    _ref_mod = sys.modules.pop(_k)     # keep the name 'libs' free for the product's compat shim

from oracle import spect_conv_oracle as O        # noqa: E402
from oracle import models_oracle as MO           # noqa: E402
from oracle.spectral_design_oracle import spectral_design as oracle_sd   # noqa: E402
from gnn_matlang_amd import synthetic            # noqa: E402


# ------------------------------------------------------------------ helpers
def ref_spectral_design(x, edge_index, **kw):
    d = Data(x=torch.tensor(np.asarray(x)), edge_index=torch.tensor(np.asarray(edge_index), dtype=torch.int64))
    d = RefSpectralDesign(nmax=0, **kw)(d)
    return dict(x=d.x.numpy(), edge_index2=d.edge_index2.numpy(), edge_attr2=d.edge_attr2.numpy(),
                lmax=np.float32(d.lmax))


def collate(graphs):
    """graphs: list of dict(x, edge_index, edge_index2, edge_attr2, y) -> one block-diagonal batch."""
    xs, e1, e2, ea, bt, ys, off = [], [], [], [], [], [], 0
    for g, d in enumerate(graphs):
        n = d['x'].shape[0]
        xs.append(d['x']); e1.append(d['edge_index'] + off); e2.append(d['edge_index2'] + off)
        ea.append(d['edge_attr2']); bt.append(np.full(n, g, dtype=np.int64)); ys.append(d['y'])
        off += n
    return dict(x=np.concatenate(xs).astype(np.float32), edge_index=np.concatenate(e1, 1),
                edge_index2=np.concatenate(e2, 1), edge_attr2=np.concatenate(ea).astype(np.float32),
                batch=np.concatenate(bt), y=np.asarray(ys, dtype=np.float32))


def design_all(raw, **kw):
    out = []
    for x, ei, y in raw:
        d = ref_spectral_design(x, ei, **kw)
        d['edge_index'] = ei
        d['y'] = y
        out.append(d)
    return out


def T(a):
    return torch.tensor(np.asarray(a))


SD_CFG = dict(
    zinc=dict(recfield=2, dv=2, nfreq=7),                                                   # Zinc12k.py:12
    counting=dict(recfield=1, dv=1, nfreq=10, adddegree=True, laplacien=False, addadj=True),  # counting.py:16
    sr25=dict(recfield=1, dv=2, nfreq=5, adddegree=True),                                   # sr25.py:16
    mutag=dict(recfield=1, dv=4, nfreq=3, adddegree=True),                                  # mutag.py:14
    mnist=dict(recfield=3, dv=10, nfreq=5),                                # prepareMnist_gnnml3_tf.py:14-17
)


# ------------------------------------------------------------------ raw data fixtures
def load_mutag():
    import scipy.io as sio
    a = sio.loadmat(os.path.join(REF, 'dataset/mutag/raw/mutag.mat'))
    A, Fm = a['A'][0], a['F'][0]
    Y = ((a['y'] + 1) // 2).astype(np.float32)                     # libs/utils.py:202
    graphs = []
    for i in range(len(A)):
        E = np.where(A[i] > 0)
        graphs.append((np.asarray(Fm[i], dtype=np.float32), np.vstack((E[0], E[1])).astype(np.int64),
                       np.float32(Y[i].item())))
    tr = np.loadtxt(os.path.join(REF, 'dataset/mutag/raw/10fold_idx/train_idx-1.txt')).astype(np.int64)
    ts = np.loadtxt(os.path.join(REF, 'dataset/mutag/raw/10fold_idx/test_idx-1.txt')).astype(np.int64)
    return graphs, tr, ts


def load_sr25():
    import networkx as nx
    gs = nx.read_graph6(os.path.join(REF, 'dataset/sr25/raw/sr251256.g6'))
    graphs = []
    for g in gs:
        n = g.number_of_nodes()
        A = np.zeros((n, n), dtype=np.float32)
        for u, v in g.edges():
            A[u, v] = A[v, u] = 1
        E = np.where(A > 0)                      # == coalesced to_undirected order (libs/utils.py:512)
        graphs.append((np.ones((n, 1), dtype=np.float32), np.vstack((E[0], E[1])).astype(np.int64), np.float32(0)))
    return graphs


def pack_graphs(graphs):
    """ragged list -> flat arrays (x_all, node_ptr, edge_all, edge_ptr, y)."""
    nptr = np.cumsum([0] + [g[0].shape[0] for g in graphs]).astype(np.int64)
    eptr = np.cumsum([0] + [g[1].shape[1] for g in graphs]).astype(np.int64)
    return dict(x=np.concatenate([g[0] for g in graphs]).astype(np.float32), node_ptr=nptr,
                edge_index=np.concatenate([g[1] for g in graphs], 1).astype(np.int64), edge_ptr=eptr,
                y=np.asarray([g[2] for g in graphs], dtype=np.float32))


# ------------------------------------------------------------------ G1 SpectralDesign
def gen_spectral_design(mutag, sr25):
    rng = np.random.default_rng(7)
    c5 = (np.ones((5, 1), np.float32),
          np.array([[0, 0, 1, 1, 2, 2, 3, 3, 4, 4], [1, 4, 0, 2, 1, 3, 2, 4, 0, 3]], dtype=np.int64), 0)
    iso = (np.ones((6, 1), np.float32), np.array([[0, 1, 1, 2, 4, 5], [1, 0, 2, 1, 5, 4]], dtype=np.int64), 0)  # node 3 isolated
    zg = synthetic.zinc_like_graph(rng)
    cg = synthetic.counting_like_graph(rng)
    mg = synthetic.mnist75_like_graph(rng)
    cases = []
    for name, g in [('c5', c5), ('iso', iso), ('zinc_like', zg), ('counting_like', cg), ('sr25_0', sr25[0]),
                    ('sr25_1', sr25[1]), ('sr25_2', sr25[2])] + [('mutag_%d' % i, mutag[i]) for i in range(5)]:
        for cfg in ('zinc', 'counting', 'sr25', 'mutag'):
            cases.append((name, cfg, g))
    cases.append(('mnist_like', 'mnist', mg))
    cases.append(('zinc_like', 'mnist', zg))          # recfield=3 on a sparse graph pins D8 (M squared twice)
    cases.append(('c5_vmax', dict(recfield=1, dv=5, nfreq=5, vmax=2.0), c5))
    cases.append(('c5_adj0', dict(recfield=0, dv=5, nfreq=3, laplacien=False), c5))
    out = {}
    for k, (name, cfg, g) in enumerate(cases):
        kw = SD_CFG[cfg] if isinstance(cfg, str) else cfg
        r = ref_spectral_design(g[0], g[1], **kw)
        o = oracle_sd(g[0], g[1], **kw)                  # oracle must agree before we trust it
        assert np.array_equal(r['edge_index2'], o['edge_index2']), (name, cfg)
        assert np.array_equal(r['x'], o['x']), (name, cfg)
        assert np.allclose(r['edge_attr2'], o['edge_attr2'], rtol=0, atol=2e-6), (name, cfg)
        p = 'case%02d/' % k
        out[p + 'name'] = np.array('%s/%s' % (name, cfg if isinstance(cfg, str) else 'custom'))
        out[p + 'kw'] = np.array(repr(kw))
        out[p + 'in_x'], out[p + 'in_edge_index'] = g[0], g[1]
        for key, v in r.items():
            out[p + key] = v
    out['ncases'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, 'spectral_design.npz'), **out)
    print('spectral_design.npz: %d cases' % len(cases))


# ------------------------------------------------------------------ G2 SpectConv
def small_batch(kind, cfg, count, seed):
    raw = synthetic.make_graphs(kind, count, seed=seed)
    return collate(design_all(raw, **SD_CFG[cfg]))


def gen_spectconv():
    out, k = {}, 0
    zb = small_batch('zinc', 'zinc', 3, 11)           # S=8
    cb = small_batch('counting', 'counting', 2, 12)   # S=12
    batches = {8: zb, 12: cb}
    for S in (1, 3, 6, 8, 12):
        base = batches[12] if S > 8 else batches[8]
        ei = base['edge_index2']
        ea = base['edge_attr2'][:, :S].copy()
        if S == 1:
            ea = np.ones_like(ea)                       # GNNML1 use (mutag.py:253)
        N = base['x'].shape[0]
        for (fin, fout) in ((7, 5), (32, 30)):
            for selfconn in (False, True):
                for depthwise in (False, True):
                    for bias in (True, False):
                        if (fin, fout) == (32, 30) and (depthwise or bias is False) and S not in (8,):
                            continue
                        torch.manual_seed(1000 + k)
                        m = RefSpectConv(fin, fout, S, selfconn=selfconn, depthwise=depthwise, bias=bias)
                        with torch.no_grad():
                            if bias:
                                m.bias.uniform_(-0.5, 0.5)
                            if depthwise:
                                m.DSweight.uniform_(-0.5, 0.5)
                        x = torch.randn(N, fin, requires_grad=True)
                        eat = T(ea).clone().requires_grad_(True)
                        y = m(x, T(ei), eat)
                        gout = torch.randn_like(y)
                        (y * gout).sum().backward()
                        # oracle agreement (bit-exact forward)
                        yo = O.spectconv_forward(x.detach(), T(ei), T(ea), m.weight.detach(),
                                                 None if m.bias is None else m.bias.detach(), selfconn, depthwise,
                                                 m.DSweight.detach() if depthwise else None)
                        assert torch.equal(yo, y.detach()), ('spectconv oracle mismatch', S, fin, selfconn, depthwise)
                        p = 'case%03d/' % k
                        out[p + 'meta'] = np.array([S, fin, fout, int(selfconn), int(depthwise), int(bias)], dtype=np.int64)
                        out[p + 'x'], out[p + 'edge_index'], out[p + 'edge_attr'] = x.detach().numpy(), ei, ea
                        out[p + 'weight'] = m.weight.detach().numpy()
                        if bias:
                            out[p + 'bias'] = m.bias.detach().numpy()
                            out[p + 'g_bias'] = m.bias.grad.numpy()
                        if depthwise:
                            out[p + 'DSweight'] = m.DSweight.detach().numpy()
                            out[p + 'g_DSweight'] = m.DSweight.grad.numpy()
                        out[p + 'out'], out[p + 'gout'] = y.detach().numpy(), gout.numpy()
                        out[p + 'g_x'], out[p + 'g_edge_attr'] = x.grad.numpy(), eat.grad.numpy()
                        out[p + 'g_weight'] = m.weight.grad.numpy()
                        k += 1
    # SpectConCatConv (spect_conv.py:105-165)
    nc = 0
    for selfconn in (True, False):
        torch.manual_seed(2000 + nc)
        S, fin, fout = 8, 9, 6
        m = RefSpectConCatConv(fin, fout, S, selfconn=selfconn)
        with torch.no_grad():
            m.bias.uniform_(-0.5, 0.5)
        x = torch.randn(zb['x'].shape[0], fin, requires_grad=True)
        eat = T(zb['edge_attr2']).clone().requires_grad_(True)
        y = m(x, T(zb['edge_index2']), eat)
        gout = torch.randn_like(y)
        (y * gout).sum().backward()
        yo = O.spectconcat_forward(x.detach(), T(zb['edge_index2']), T(zb['edge_attr2']), m.weight.detach(),
                                   m.bias.detach(), selfconn)
        assert torch.equal(yo, y.detach())
        p = 'concat%d/' % nc
        out[p + 'meta'] = np.array([S, fin, fout, int(selfconn)], dtype=np.int64)
        out[p + 'x'], out[p + 'edge_index'], out[p + 'edge_attr'] = x.detach().numpy(), zb['edge_index2'], zb['edge_attr2']
        out[p + 'weight'], out[p + 'bias'] = m.weight.detach().numpy(), m.bias.detach().numpy()
        out[p + 'out'], out[p + 'gout'] = y.detach().numpy(), gout.numpy()
        out[p + 'g_x'], out[p + 'g_edge_attr'] = x.grad.numpy(), eat.grad.numpy()
        out[p + 'g_weight'], out[p + 'g_bias'] = m.weight.grad.numpy(), m.bias.grad.numpy()
        nc += 1
    out['ncases'], out['nconcat'] = np.int64(k), np.int64(nc)
    np.savez_compressed(os.path.join(OUT, 'spectconv.npz'), **out)
    print('spectconv.npz: %d + %d cases' % (k, nc))


# ------------------------------------------------------------------ G3 ML3Layer
def gen_ml3layer():
    out, k = {}, 0
    zb = small_batch('zinc', 'zinc', 3, 21)
    cb = small_batch('counting', 'counting', 2, 22)
    sb = collate(design_all(load_sr25()[:2], **SD_CFG['sr25']))
    for bname, b, (ninp, nout1, nout2s) in (('zinc', zb, (25, 30, (2, 0))), ('zinc32', zb, (32, 30, (2,))),
                                            ('counting', cb, (2, 16, (16,))), ('counting32', cb, (32, 16, (16,))),
                                            ('sr25', sb, (2, 32, (16,))), ('sr25_48', sb, (48, 32, (16, 0)))):
        ne = b['edge_attr2'].shape[1]
        for learnedge in (True, False):
            for nout2 in nout2s:
                torch.manual_seed(3000 + k)
                m = RefML3Layer(learnedge, ne, ne, ninp, nout1, nout2)
                with torch.no_grad():
                    m.conv1.bias.uniform_(-0.3, 0.3)
                if ninp == b['x'].shape[1]:
                    x = T(b['x']).clone()
                else:
                    x = torch.randn(b['x'].shape[0], ninp)
                x.requires_grad_(True)
                eat = T(b['edge_attr2']).clone().requires_grad_(True)
                y = m(x, T(b['edge_index2']), eat)
                gout = torch.randn_like(y)
                (y * gout).sum().backward()
                params = {n: p.detach() for n, p in m.named_parameters()}
                yo = O.ml3layer_forward(x.detach(), T(b['edge_index2']), T(b['edge_attr2']), params, learnedge, nout2)
                assert torch.equal(yo, y.detach()), ('ml3 oracle mismatch', bname, learnedge, nout2)
                p = 'case%03d/' % k
                out[p + 'meta'] = np.array([int(learnedge), ne, ne, ninp, nout1, nout2], dtype=np.int64)
                out[p + 'x'], out[p + 'edge_index'], out[p + 'edge_attr'] = x.detach().numpy(), b['edge_index2'], b['edge_attr2']
                for n, prm in m.named_parameters():
                    out[p + 'param/' + n] = prm.detach().numpy()
                    out[p + 'grad/' + n] = prm.grad.numpy()
                out[p + 'out'], out[p + 'gout'] = y.detach().numpy(), gout.numpy()
                out[p + 'g_x'], out[p + 'g_edge_attr'] = x.grad.numpy(), eat.grad.numpy()
                k += 1
    out['ncases'] = np.int64(k)
    np.savez_compressed(os.path.join(OUT, 'ml3layer.npz'), **out)
    print('ml3layer.npz: %d cases' % k)


# ------------------------------------------------------------------ G4 models
def run_model_fixture(fname, model, batch, call, loss_fn, lr, train_mode=True, nsteps=5):
    """one batch: logits, loss, all grads after one backward; then nsteps Adam steps' losses."""
    out = {}
    for k in ('x', 'edge_index', 'edge_index2', 'edge_attr2', 'batch', 'y'):
        out['batch/' + k] = batch[k]
    for n, p in model.state_dict().items():
        out['param/' + n] = p.detach().numpy().copy()
    model.train(train_mode)
    pre = call(model)
    loss = loss_fn(pre, T(batch['y']))
    model.zero_grad()
    loss.backward()
    out['logits'], out['loss'] = pre.detach().numpy(), np.float32(loss.item())
    for n, p in model.named_parameters():
        out['grad/' + n] = p.grad.numpy().copy()
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    traj = []
    for _ in range(nsteps):
        opt.zero_grad()
        l = loss_fn(call(model), T(batch['y']))
        l.backward()
        opt.step()
        traj.append(l.item())
    out['loss_traj'] = np.asarray(traj, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, 'loss', out['loss'], 'traj', traj)


def gen_models(mutag, mutag_tr, sr25):
    # H3 ZINC GNNML3 (Zinc12k.py:310-371): synthetic ZINC-like batch of 8 graphs
    zb = small_batch('zinc', 'zinc', 8, 31)
    torch.manual_seed(0)
    m = MO.zinc_gnnml3(ninp=25, ne=8, layer_cls=RefML3Layer)
    g = lambda mm, b=zb: mm(T(b['x']), T(b['edge_index2']), T(b['edge_attr2']), T(b['batch']), len(b['y']))
    run_model_fixture('model_zinc_gnnml3.npz', m, zb, g, MO.zinc_loss, 1e-3)

    # H2 counting GNNML3 (counting.py:335-416): bs 10, x=[1,deg/max], y=tri/std
    cb = small_batch('counting', 'counting', 10, 32)
    cb['x'][:, 1] = cb['x'][:, 1] / cb['x'][:, 1].max()                  # counting.py:22
    cb['y'] = (cb['y'] / cb['y'].std()).astype(np.float32)               # counting.py:20
    torch.manual_seed(0)
    m = MO.counting_gnnml3(ninp=2, ne=12, layer_cls=RefML3Layer)
    g = lambda mm, b=cb: mm(T(b['x']), T(b['edge_index2']), T(b['edge_attr2']), T(b['batch']), len(b['y']))
    run_model_fixture('model_counting_gnnml3.npz', m, cb, g, MO.counting_loss, 1e-3)

    # H1 mutag (mutag.py:214-359): fold-1 first 16 train ids in file order, real data
    mb = collate(design_all([mutag[i] for i in mutag_tr[:16]], **SD_CFG['mutag']))
    torch.manual_seed(0)
    m = MO.mutag_gnnml3(ninp=8, ne=4, layer_cls=RefML3Layer)
    g = lambda mm, b=mb: mm(T(b['x']), T(b['edge_index2']), T(b['edge_attr2']), T(b['batch']), len(b['y']))
    run_model_fixture('model_mutag_gnnml3.npz', m, mb, g, MO.mutag_loss, 1e-3)
    torch.manual_seed(0)
    m = MO.OracleGNNML1Mutag(ninp=8, conv_cls=RefSpectConv)
    g = lambda mm, b=mb: mm(T(b['x']), T(b['edge_index']), T(b['batch']), len(b['y']))
    run_model_fixture('model_mutag_gnnml1.npz', m, mb, g, MO.mutag_loss, 1e-3)

    # H5 sr25 GNNML3 (sr25.py:248-300): all 15 graphs, forward-only, seeds 0..2
    sb = collate(design_all(sr25, **SD_CFG['sr25']))
    out = {('batch/' + k): sb[k] for k in ('x', 'edge_index', 'edge_index2', 'edge_attr2', 'batch', 'y')}
    Mcnt = 0
    for seed in range(3):
        torch.manual_seed(seed)
        m = MO.sr25_gnnml3(ninp=2, ne=6, layer_cls=RefML3Layer)
        m.eval()
        with torch.no_grad():
            E = m(T(sb['x']), T(sb['edge_index2']), T(sb['edge_attr2']), T(sb['batch']), 15).numpy()
        for n, p in m.state_dict().items():
            out['seed%d/param/%s' % (seed, n)] = p.numpy().copy()
        out['seed%d/emb' % seed] = E
        Mcnt = Mcnt + 1 * ((np.abs(np.expand_dims(E, 1) - np.expand_dims(E, 0))).sum(2) > 0.001)   # sr25.py:298
        out['seed%d/similar' % seed] = np.int64(((Mcnt == 0).sum() - Mcnt.shape[0]) / 2)          # sr25.py:299
    np.savez_compressed(os.path.join(OUT, 'model_sr25_gnnml3.npz'), **out)
    print('model_sr25_gnnml3.npz similar:', [int(out['seed%d/similar' % s]) for s in range(3)])
    gen_sr25_gnnml1(sb)


def gen_sr25_gnnml1(sb):
    """sr25.py:192-246,282-300 with model = GNNML1() (the sum form, tanh): embeddings of the 15 graphs and the running
    count of never-separated pairs for seeds 0..2; the reference SpectConv class is injected for conv_i1."""
    out = {('batch/' + k): sb[k] for k in ('x', 'edge_index', 'edge_index2', 'edge_attr2', 'batch', 'y')}
    Mcnt = 0
    for seed in range(3):
        torch.manual_seed(seed)
        m = MO.OracleGNNML1Sum(ninp=2, conv_cls=RefSpectConv)
        m.eval()
        with torch.no_grad():
            E = m(T(sb['x']), T(sb['edge_index']), T(sb['batch']), 15).numpy()
        for n, p in m.state_dict().items():
            out['seed%d/param/%s' % (seed, n)] = p.numpy().copy()
        out['seed%d/emb' % seed] = E
        Mcnt = Mcnt + 1 * ((np.abs(np.expand_dims(E, 1) - np.expand_dims(E, 0))).sum(2) > 0.001)   # sr25.py:298
        out['seed%d/similar' % seed] = np.int64(((Mcnt == 0).sum() - Mcnt.shape[0]) / 2)          # sr25.py:299
    np.savez_compressed(os.path.join(OUT, 'model_sr25_gnnml1.npz'), **out)
    print('model_sr25_gnnml1.npz similar:', [int(out['seed%d/similar' % s]) for s in range(3)])


# ------------------------------------------------------------------ H4: MNIST-75 GNNML3 of the TF pipeline
def gen_mnist_tf():
    """Config 4 exists only as TensorFlow-1.15 code (libs/models_tf.py:223-268 DSGCNN, libs/layers_tf.py:193-245
    GraphConvolutionBatch, :323-351 ReadoutLayer, :85-135 Dense, libs/metrics_tf.py:17-20) and TensorFlow is not
    installed, so the fixture restates that GRAPH -- dense-batched, exactly as the TF code writes it -- in float32 torch:

        per layer   out = relu( add_n_i tensordot( matmul(support[:, i], x), W_i ) + bias )       layers_tf.py:231-241
        readout     batch_normalization( reduce_sum(x, 1) / ND )   (training=True, epsilon 1e-3)  layers_tf.py:343-349
        head        relu(x W + b) ; x W + b                                                        layers_tf.py:117-128
        loss        reduce_mean( softmax_cross_entropy_with_logits )                               metrics_tf.py:17-20
        optimiser   tf.train.AdamOptimizer(0.01): lr_t = lr sqrt(1-b2^t)/(1-b1^t); p -= lr_t m / (sqrt(v) + 1e-8)

    The supports [B, 6, 75, 75] are laid out as prepareMnist_gnnml3_tf.py:33-66 does (5 Gaussians with dv = 10 on the
    (A+I)^4 > 0 mask + identity) from the output of the reference's own SpectralDesign (same formulas, libs/utils.py:
    546-610, run unmodified).  This formulation shares nothing with the sparse gather / scatter one of the oracle and the
    kernels except the inputs.  dropout: the placeholder's default 0."""
    raw = synthetic.make_graphs('mnist75', 6, seed=9)
    gs = design_all(raw, **SD_CFG['mnist'])
    b = collate(gs)
    B, n, S = len(gs), 75, 6
    SP = np.zeros((B, S, n, n), dtype=np.float32)
    for i, d in enumerate(gs):
        e0, e1 = d['edge_index2']
        SP[i, :, e0, e1] = d['edge_attr2']                          # SP[:, E0, E1] = edge_attr2^T   (utils.py:608-610)
    X = torch.tensor(b['x'].reshape(B, n, 2))
    Y = torch.nn.functional.one_hot(torch.tensor(b['y'].astype(np.int64)), 10).float()
    SPt = torch.tensor(SP)
    ND = torch.full((B, 1), 75.0)
    torch.manual_seed(5)
    dims = [(2, 64), (64, 128), (128, 128)]

    def glorot(shape):                                                # inits_tf.py:12-16
        r = float(np.sqrt(6.0 / (shape[0] + shape[1])))
        return torch.nn.Parameter((torch.rand(shape) * 2 - 1) * r)
    Wc = [[glorot(d) for _ in range(S)] for d in dims]
    bc = [torch.nn.Parameter(torch.randn(d[1]) * 0.05) for d in dims]   # (zeros in the reference; non-zero exercises the path)
    gamma, beta = torch.nn.Parameter(torch.ones(128) + 0.1 * torch.randn(128)), torch.nn.Parameter(0.1 * torch.randn(128))
    W1, b1 = glorot((128, 32)), torch.nn.Parameter(torch.randn(32) * 0.05)
    W2, b2 = glorot((32, 10)), torch.nn.Parameter(torch.randn(10) * 0.05)
    params = [w for ws in Wc for w in ws] + bc + [gamma, beta, W1, b1, W2, b2]

    def forward():
        x = X
        for l in range(3):
            outs = [torch.tensordot(torch.matmul(SPt[:, i], x), Wc[l][i], dims=([2], [0])) for i in range(S)]
            x = torch.relu(sum(outs) + bc[l])
        o = x.sum(1) / ND
        mean, var = o.mean(0), o.var(0, unbiased=False)
        o = (o - mean) / torch.sqrt(var + 1e-3) * gamma + beta
        o = torch.relu(o @ W1 + b1)
        return o @ W2 + b2

    def loss_of(logits):
        return (-(Y * torch.log_softmax(logits, 1)).sum(1)).mean()

    def state():                                                      # under the names of gnn_matlang_amd.models.mnist_gnnml3
        sd = {}
        for l in range(3):
            sd['conv%d.conv1.weight' % (l + 1)] = torch.stack([w.detach() for w in Wc[l]]).numpy().copy()
            sd['conv%d.conv1.bias' % (l + 1)] = bc[l].detach().numpy().copy()
        sd['bnr.weight'], sd['bnr.bias'] = gamma.detach().numpy().copy(), beta.detach().numpy().copy()
        sd['fc1.weight'], sd['fc1.bias'] = W1.detach().t().numpy().copy(), b1.detach().numpy().copy()
        sd['fc2.weight'], sd['fc2.bias'] = W2.detach().t().numpy().copy(), b2.detach().numpy().copy()
        return sd

    def grads():
        g = {}
        for l in range(3):
            g['conv%d.conv1.weight' % (l + 1)] = torch.stack([w.grad for w in Wc[l]]).numpy().copy()
            g['conv%d.conv1.bias' % (l + 1)] = bc[l].grad.numpy().copy()
        g['bnr.weight'], g['bnr.bias'] = gamma.grad.numpy().copy(), beta.grad.numpy().copy()
        g['fc1.weight'], g['fc1.bias'] = W1.grad.t().numpy().copy(), b1.grad.numpy().copy()
        g['fc2.weight'], g['fc2.bias'] = W2.grad.t().numpy().copy(), b2.grad.numpy().copy()
        return g
    out = {('batch/' + k): b[k] for k in ('x', 'edge_index', 'edge_index2', 'edge_attr2', 'batch')}
    out['batch/y'] = b['y'].astype(np.int64)
    for k, v in state().items():
        out['param/' + k] = v
    logits = forward()
    loss = loss_of(logits)
    loss.backward()
    out['logits'], out['loss'] = logits.detach().numpy(), np.float32(loss.item())
    for k, v in grads().items():
        out['grad/' + k] = v
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    traj, lr, b1_, b2_ = [], 0.01, 0.9, 0.999
    for t in range(1, 6):                                             # tf.train.AdamOptimizer's update rule
        for p_ in params:
            p_.grad = None
        l = loss_of(forward())
        l.backward()
        traj.append(l.item())
        lr_t = lr * np.sqrt(1 - b2_ ** t) / (1 - b1_ ** t)
        with torch.no_grad():
            for i, p_ in enumerate(params):
                m[i] = b1_ * m[i] + (1 - b1_) * p_.grad
                v[i] = b2_ * v[i] + (1 - b2_) * p_.grad * p_.grad
                p_ -= lr_t * m[i] / (torch.sqrt(v[i]) + 1e-8)
    out['loss_traj'] = np.asarray(traj, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, 'model_mnist_gnnml3_tf.npz'), **out)
    print('model_mnist_gnnml3_tf.npz loss', out['loss'], 'traj', traj)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)          # deterministic summation order inside matmul
    mutag, tr, ts = load_mutag()
    sr25 = load_sr25()
    d = pack_graphs(mutag)
    d['train_idx_fold1'], d['test_idx_fold1'] = tr, ts
    np.savez_compressed(os.path.join(OUT, 'data_mutag.npz'), **d)
    np.savez_compressed(os.path.join(OUT, 'data_sr25.npz'), **pack_graphs(sr25))
    gen_spectral_design(mutag, sr25)
    gen_spectconv()
    gen_ml3layer()
    gen_models(mutag, tr, sr25)
    gen_mnist_tf()


if __name__ == '__main__':
    main()
