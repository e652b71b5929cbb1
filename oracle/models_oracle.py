"""Oracle model assemblies (CPU) for the BASELINE configs' callers of the hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  The reference defines its models
inside experiment scripts that cannot be imported (they load datasets at import
time), so the thin assembly around the layers -- stack, pooling, head -- is
restated here from the cited lines, with the LAYER classes injectable: the
oracle's own layers by default, the reference's imported ``ML3Layer`` /
``SpectConv`` when ``oracle/make_golden.py`` generates fixtures.  Attribute names
equal the reference's so ``state_dict`` keys line up.

    GNNML3  : Zinc12k.py:310-345 (4 layers 30+2, add-pool, fc 32->1)
              counting.py:335-372 (5 layers 16+16, add-pool)
              sr25.py:248-278     (3 layers 32+16, add-pool, tanh(fc nin->10))
              mutag.py:268-309    (3 layers 24+24, learnedge=False, BatchNorm, mean-pool)
    GNNML1  : mutag.py:214-266    (S=1 unit supports, concat, BatchNorm, mean-pool)
"""
import torch
import torch.nn.functional as F

from .spect_conv_oracle import OracleML3Layer, OracleSpectConv


def global_add_pool(x, batch, num_graphs):
    out = torch.zeros(num_graphs, x.size(1), dtype=x.dtype, device=x.device)
    return out.index_add_(0, batch, x)


def global_mean_pool(x, batch, num_graphs):
    cnt = torch.zeros(num_graphs, dtype=x.dtype, device=x.device)
    cnt.index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return global_add_pool(x, batch, num_graphs) / cnt.clamp(min=1).unsqueeze(-1)


class OracleGNNML3(torch.nn.Module):
    """``head``: 'mlp32' -> fc2(relu(fc1)) with fc1: nin->32, fc2: 32->nclass;
                 'tanh10' -> tanh(fc1) with fc1: nin->10 (sr25.py:262,276)."""

    def __init__(self, ninp, ne, nout1, nout2, nlayers, learnedge=True, bn=False,
                 pool='add', head='mlp32', nclass=1, readout_bn=False, layer_cls=OracleML3Layer):
        super().__init__()
        widths = list(nout1) if isinstance(nout1, (list, tuple)) else [nout1] * nlayers    # per-layer nout1
        self.nlayers, self.bn, self.pool, self.head, self.readout_bn = nlayers, bn, pool, head, readout_bn
        fin = ninp
        for i in range(nlayers):
            setattr(self, 'conv%d' % (i + 1),
                    layer_cls(learnedge=learnedge, nedgeinput=ne, nedgeoutput=ne,
                              ninp=fin, nout1=widths[i], nout2=nout2))
            fin = widths[i] + nout2
            if bn:
                setattr(self, 'bn%d' % (i + 1), torch.nn.BatchNorm1d(fin))
        nin = fin
        if readout_bn:                       # TF ReadoutLayer: batch_normalization of the pooled vector
            # tf.layers.batch_normalization defaults (libs/layers_tf.py:349): epsilon 1e-3, momentum 0.99 (= 0.01 in torch's convention)
            self.bnr = torch.nn.BatchNorm1d(nin, eps=1e-3, momentum=0.01)
        if head == 'mlp32':
            self.fc1 = torch.nn.Linear(nin, 32)
            self.fc2 = torch.nn.Linear(32, nclass)
        else:
            self.fc1 = torch.nn.Linear(nin, 10)

    def forward(self, x, edge_index2, edge_attr2, batch, num_graphs):
        for i in range(self.nlayers):
            x = getattr(self, 'conv%d' % (i + 1))(x, edge_index2, edge_attr2)
            if self.bn:
                x = getattr(self, 'bn%d' % (i + 1))(x)
        pool = global_add_pool if self.pool == 'add' else global_mean_pool
        x = pool(x, batch, num_graphs)
        if self.readout_bn:
            x = self.bnr(x)
        if self.head == 'mlp32':
            return self.fc2(F.relu(self.fc1(x)))
        return torch.tanh(self.fc1(x))


class OracleGNNML1Mutag(torch.nn.Module):
    """mutag.py:214-266.  conv*1 are SpectConv(K=1, selfconn=False) over the RAW
    adjacency with unit edge values (:253; the hard-coded .to('cuda') of the
    reference is SURVEY D4 -- the values are what matters)."""

    def __init__(self, ninp, nout1=16, nout2=32, nout3=16, conv_cls=OracleSpectConv):
        super().__init__()
        nin = nout1 + nout2 + nout3
        for i, fin in enumerate([ninp, nin, nin], start=1):
            setattr(self, 'bn%d' % i, torch.nn.BatchNorm1d(nin))
            setattr(self, 'conv%d1' % i, conv_cls(fin, nout2, 1, selfconn=False))
            setattr(self, 'fc%d1' % i, torch.nn.Linear(fin, nout1))
            setattr(self, 'fc%d2' % i, torch.nn.Linear(fin, nout3))
            setattr(self, 'fc%d3' % i, torch.nn.Linear(fin, nout3))
        self.fc1 = torch.nn.Linear(nin, 32)
        self.fc2 = torch.nn.Linear(32, 1)

    def forward(self, x, edge_index, batch, num_graphs):
        ones = torch.ones(edge_index.shape[1], 1, dtype=x.dtype, device=x.device)
        for i in (1, 2, 3):
            g = lambda n: getattr(self, n % i)
            x = torch.cat([F.relu(g('fc%d1')(x)),
                           F.relu(g('conv%d1')(x, edge_index, ones)),
                           F.relu(g('fc%d2')(x)) * F.relu(g('fc%d3')(x))], 1)
            x = g('bn%d')(x)
        x = global_mean_pool(x, batch, num_graphs)
        return self.fc2(F.relu(self.fc1(x)))


class OracleGNNML1Sum(torch.nn.Module):
    """sr25.py:192-246 (graph8c.py:205-246 alike): x <- tanh(fc_i1 x + conv_i1 x + fc_i2 x * fc_i3 x), three blocks,
    conv_i1 = SpectConv(K=1, selfconn=False) with unit edge values, add-pool, fc1: nout -> 10."""

    def __init__(self, ninp, nout=64, conv_cls=OracleSpectConv):
        super().__init__()
        for i, fin in enumerate([ninp, nout, nout], start=1):
            setattr(self, 'conv%d1' % i, conv_cls(fin, nout, selfconn=False))
            for j in (1, 2, 3):
                setattr(self, 'fc%d%d' % (i, j), torch.nn.Linear(fin, nout))
        self.fc1 = torch.nn.Linear(nout, 10)

    def forward(self, x, edge_index, batch, num_graphs):
        ones = torch.ones(edge_index.shape[1], 1, dtype=x.dtype, device=x.device)
        for i in (1, 2, 3):
            g = lambda n: getattr(self, n % i)
            x = torch.tanh(g('fc%d1')(x) + g('conv%d1')(x, edge_index, ones) + g('fc%d2')(x) * g('fc%d3')(x))
        return self.fc1(global_add_pool(x, batch, num_graphs))


# ---- per-config constructors (shapes cited above) --------------------------
def zinc_gnnml3(ninp=25, ne=8, **kw):
    return OracleGNNML3(ninp, ne, 30, 2, 4, **kw)


def counting_gnnml3(ninp=2, ne=12, **kw):
    return OracleGNNML3(ninp, ne, 16, 16, 5, **kw)


def sr25_gnnml3(ninp=2, ne=6, **kw):
    return OracleGNNML3(ninp, ne, 32, 16, 3, head='tanh10', **kw)


def mnist_gnnml3(ninp=2, ne=6, **kw):
    """mnist75_gnnml3_tf.py:62 + libs/models_tf.py:223-268 (DSGCNN): 3 graph convolutions 64/128/128 over
    S=6 dense supports, no edge learning, no Hadamard branch, mean readout + batch norm, 128->32->10.
    (dropout of the TF model is a training-time regulariser, not part of the path; omitted.)"""
    return OracleGNNML3(ninp, ne, [64, 128, 128], 0, 3, learnedge=False, pool='mean', head='mlp32', nclass=10,
                        readout_bn=True, **kw)


def mutag_gnnml3(ninp=8, ne=4, **kw):
    return OracleGNNML3(ninp, ne, 24, 24, 3, learnedge=False, bn=True, pool='mean', **kw)


# ---- losses of the training loops ------------------------------------------
def zinc_loss(pre, y):            # Zinc12k.py:365
    return F.l1_loss(pre, y.unsqueeze(-1), reduction='sum')


def counting_loss(pre, y):        # counting.py:411 (y already the selected task column)
    return torch.square(pre - y.view(-1, 1)).sum()


def mnist_loss(pre, y):           # libs/metrics_tf.py softmax_cross_entropy (mean over the batch)
    return F.cross_entropy(pre, y.long())


def mutag_loss(pre, y):           # mutag.py:345-348
    return F.binary_cross_entropy(torch.sigmoid(pre)[:, 0], y, reduction='sum')
