"""GNNML1 / GNNML3 model assemblies of the BASELINE configs on top of the MI355X layers.

The reference defines these classes inside its experiment scripts with hard-coded sizes
(Zinc12k.py:310-345, counting.py:335-372, sr25.py:248-278, mutag.py:214-309); here they are one
configurable class each with the same attribute names, so ``state_dict`` keys match the reference's.
``forward`` takes a ``gnn_matlang_amd.Batch`` (PyG-compatible field names: x, edge_index,
edge_index2, edge_attr2, batch, ptr, y).
"""
import torch
import torch.nn.functional as F

from .functional import segment_bcast, segment_max, segment_max_bwd, segment_sum, skip_last_mask, tall_linear, _ptr32
from .spect_conv import ML3Layer, SpectConv
from .dist import SyncBatchNorm1d


class BatchNorm1d(torch.nn.BatchNorm1d):
    """torch.nn.BatchNorm1d (same parameters, buffers, state_dict keys: mutag.py:272-288) whose TRAINING pass over a float32 CUDA
    [N, C] input runs on csrc/gml_bn.hip (C <= 64, C % 4 == 0, float4-addressable rows); eval mode, other shapes and dtypes: the
    base class.  GML_TORCH_BN=1 in the environment: always the base class (A/B)."""

    def forward(self, x):
        import os
        if not (self.training or self.running_mean is None) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 \
                or x.size(1) > 64 or x.size(1) % 4 or x.stride(1) != 1 or x.stride(0) % 4 or x.stride(0) < x.size(1) or x.data_ptr() % 16 \
                or x.size(0) < 2 \
                or os.environ.get('GML_TORCH_BN'):
            return super().forward(x)
        from .functional import BatchNormFunction
        y, stats = BatchNormFunction.apply(x, self.weight, self.bias, self.eps)
        if self.training and self.running_mean is not None:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                n = x.size(0)
                self.running_mean.mul_(1 - m).add_(stats[0], alpha=m)
                self.running_var.mul_(1 - m).add_(stats[1], alpha=m * n / (n - 1))
        return y


class _SegmentPool(torch.autograd.Function):
    """global_add_pool / global_mean_pool (torch_geometric.nn, used at Zinc12k.py:343, mutag.py:307)."""

    @staticmethod
    def forward(ctx, x, ptr, batch, mean):
        ctx.save_for_backward(ptr, batch)
        ctx.mean = mean
        out = segment_sum(x.contiguous(), ptr, mean)
        if int(mean) & 2:
            skip_last_mask(out)
        return out

    @staticmethod
    def backward(ctx, g):
        ptr, batch = ctx.saved_tensors
        g = g.contiguous()
        if int(ctx.mean) & 2:          # GML_POOL_SKIP_LAST: the padding graph's row does not depend on x -- no gradient to its nodes
            g = g * skip_last_mask(g)
        return segment_bcast(g, ptr, batch.numel(), bool(int(ctx.mean) & 1)), None, None, None


class _SegmentMax(torch.autograd.Function):
    """global_max_pool (torch_geometric.nn, used at /root/reference/enzymes.py:384): the gradient goes to the arg-max row."""

    @staticmethod
    def forward(ctx, x, ptr):
        out, arg = segment_max(x.contiguous(), ptr)
        ctx.save_for_backward(ptr, arg)
        ctx.nrows = x.size(0)
        return out

    @staticmethod
    def backward(ctx, g):
        ptr, arg = ctx.saved_tensors
        return segment_max_bwd(g.contiguous(), ptr, arg, ctx.nrows), None


def global_max_pool(x, data):
    return _SegmentMax.apply(x, data.ptr)


def _pad_flag(data):
    return 2 if getattr(data, 'pad_graph', False) else 0       # GML_POOL_SKIP_LAST: the padding graph of a static-shape batch


def global_add_pool(x, data):
    return _SegmentPool.apply(x, data.ptr, data.batch, _pad_flag(data))


def global_mean_pool(x, data):
    return _SegmentPool.apply(x, data.ptr, data.batch, 1 | _pad_flag(data))


class GNNML3(torch.nn.Module):
    """head 'mlp32': fc2(relu(fc1 x)), fc1: nin->32, fc2: 32->nclass;  'tanh10': tanh(fc1 x), fc1: nin->10."""

    def __init__(self, ninp, ne, nout1, nout2, nlayers, learnedge=True, bn=False, pool='add', head='mlp32',
                 nclass=1, readout_bn=False, dense_n=0, chain=True):
        super().__init__()
        # chain=False: do not declare the relu hand-over between stacked layers (ML3Layer.chain_after).  The hand-over is
        # valid only while convN+1 is the SOLE consumer of convN's output tensor: a forward hook, an auxiliary loss or a
        # feature tap on an intermediate output adds a second gradient the lower layer would then leave unmasked -- pass
        # chain=False for such models (GML_NO_CHAIN=1 switches it off process-wide).
        # dense_n > 0: equal-size graphs of dense_n nodes with near-dense masks (MNIST-75) are evaluated as dense
        # blocks (dense_block.py: HIP batched support product + one tall GEMM per layer); same parameters, same values
        if dense_n and (learnedge or nout2):
            raise ValueError('the dense-block path covers plain SpectConv stacks (learnedge=False, nout2=0)')
        self.dense_n = int(dense_n)
        widths = list(nout1) if isinstance(nout1, (list, tuple)) else [nout1] * nlayers    # per-layer nout1
        self.nlayers, self.bn, self.pool, self.head, self.readout_bn = nlayers, bn, pool, head, readout_bn
        fin = ninp
        for i in range(nlayers):
            setattr(self, 'conv%d' % (i + 1),
                    ML3Layer(learnedge=learnedge, nedgeinput=ne, nedgeoutput=ne,
                             ninp=fin, nout1=widths[i], nout2=nout2))
            fin = widths[i] + nout2
            if bn:
                setattr(self, 'bn%d' % (i + 1), BatchNorm1d(fin))
        if chain and not bn and not self.dense_n:
            # x = conv2(conv1(x, ...), ...) with nothing else reading the intermediate outputs (Zinc12k.py:338-341,
            # counting.py:361-366, sr25.py:266-270): the relu hand-over between stacked layers applies
            for i in range(1, nlayers):
                getattr(self, 'conv%d' % (i + 1)).chain_after(getattr(self, 'conv%d' % i))
        nin = fin
        if readout_bn:                       # TF ReadoutLayer: batch_normalization of the pooled vector
            # tf.layers.batch_normalization defaults (libs/layers_tf.py:349): epsilon 1e-3, momentum 0.99 (= 0.01 in torch's convention)
            self.bnr = SyncBatchNorm1d(nin, eps=1e-3, momentum=0.01)   # global-batch statistics under data parallelism (dist.py)
        if head == 'mlp32':
            self.fc1 = torch.nn.Linear(nin, 32)
            self.fc2 = torch.nn.Linear(32, nclass)
        else:
            self.fc1 = torch.nn.Linear(nin, 10)

    def forward(self, data, _features=False, _capture=None, _pad_grad_zero=False):
        x = data.x
        if getattr(data, 'pad_graph', False) and self.training and (self.bn or self.readout_bn):
            # the padding nodes / the padding graph's pooled row would enter the batch statistics
            raise NotImplementedError('BatchNorm models take plain batches: a padded static batch (pad_graph) would count its padding '
                                      'rows in the statistics')
        if self.dense_n:
            from .dense_block import dense_supports, spectconv_dense, _library
            sp = getattr(data, '_spT', None)                    # per-batch data, like the CSR
            if sp is None or (sp.blocks is not None) != bool(_library()):
                data._spT = dense_supports(data.edge_index2, data.edge_attr2, data.ptr, self.dense_n)
        else:
            csr = data.csr('edge_index2')
        pooled = False
        for i in range(self.nlayers):
            layer = getattr(self, 'conv%d' % (i + 1))
            if self.dense_n:
                x = spectconv_dense(x, data._spT, layer.conv1.weight, layer.conv1.bias, self.dense_n, relu=True)
            elif i == self.nlayers - 1 and not self.bn and self.pool in ('add', 'mean') and torch.is_grad_enabled():
                # the pool directly follows the last layer: one autograd node, the pool's gradient is not expanded to [N, C]
                if getattr(data, '_batch_i32', None) is None:
                    data._batch_i32 = data.batch.to(torch.int32).contiguous()
                x = layer.forward_pooled(x, csr, data.edge_attr2, _ptr32(data.ptr), data._batch_i32,
                                         int(self.pool == 'mean') | (2 if getattr(data, 'pad_graph', False) else 0) | (4 if _pad_grad_zero else 0))   # 2: GML_POOL_SKIP_LAST; 4: its gradient is zero already
                pooled = True
            elif self.bn and i == 0:
                # BatchNorm layers behind the first layer amplify the ~1e-5 relative error of the split bf16 products in ITS weight /
                # bias gradients past 1e-4 of their scale (mutag.py:272-288: 1.6e-4 measured): that one layer runs on the exact
                # f32-product kernels, forward and backward (the narrowest layer of the stack: 8 -> 48 features)
                from . import functional as _Fn
                with _Fn.exact_products():
                    x = layer(x, csr, data.edge_attr2)
            else:
                x = layer(x, csr, data.edge_attr2)
            if self.bn:
                x = getattr(self, 'bn%d' % (i + 1))(x)
        if not pooled:
            x = {'add': global_add_pool, 'mean': global_mean_pool, 'max': global_max_pool}[self.pool](x, data)
        if self.readout_bn:
            x = self.bnr(x)
        if _features:
            return x
        if self.head == 'mlp32':
            z1 = tall_linear(x, self.fc1)
            if _capture is not None:
                _capture['head_pre'] = z1.detach()            # (the parity checker takes the head's relu mask from here)
            return tall_linear(F.relu(z1), self.fc2)
        return torch.tanh(tall_linear(x, self.fc1))

    def features(self, data, pad_grad_zero=False):
        """the pooled (and, with readout_bn, normalised) graph features the head is applied to: [num_graphs, nin].
        pad_grad_zero: the caller's loss gives the padding graph's row (static batches) a zero gradient -- no masking launch."""
        return self.forward(data, _features=True, _pad_grad_zero=pad_grad_zero)


def _gnnml1_block(x, csr, fc1, conv, fc2, fc3, mode, act):
    """the block as one fused launch (functional.GNNML1BlockFunction) or None when the widths are outside the kernel (> 64) or the
    conv is not the plain K = 1 form the scripts use.  Unit edge values (sr25.py:231, mutag.py:253: torch.ones)."""
    from . import functional as Fn
    if conv.weight.size(0) != 1 or conv.selfconn or conv.depthwise:
        return None
    Fin, n1, n2, n3 = int(x.size(1)), int(fc1.weight.size(0)), int(conv.weight.size(2)), int(fc2.weight.size(0))
    if not Fn.gnnml1_block_supported(x, Fin, n1, n2, n3, mode):
        return None
    return Fn.GNNML1BlockFunction.apply(x, csr, None, fc1.weight, fc1.bias, conv.weight, conv.bias, fc2.weight, fc2.bias,
                                        fc3.weight, fc3.bias, mode, act)


class GNNML1Mutag(torch.nn.Module):
    """mutag.py:214-266: three blocks of [relu(fc x) | relu(SpectConv_{S=1}(x)) | relu(fc x)*relu(fc x)] + BN."""

    def __init__(self, ninp, nout1=16, nout2=32, nout3=16):
        super().__init__()
        nin = nout1 + nout2 + nout3
        for i, fin in enumerate([ninp, nin, nin], start=1):
            setattr(self, 'bn%d' % i, BatchNorm1d(nin))
            setattr(self, 'conv%d1' % i, SpectConv(fin, nout2, 1, selfconn=False))
            setattr(self, 'fc%d1' % i, torch.nn.Linear(fin, nout1))
            setattr(self, 'fc%d2' % i, torch.nn.Linear(fin, nout3))
            setattr(self, 'fc%d3' % i, torch.nn.Linear(fin, nout3))
        self.fc1 = torch.nn.Linear(nin, 32)
        self.fc2 = torch.nn.Linear(32, 1)

    def forward(self, data):
        x = data.x
        csr = data.csr('edge_index')
        ones = torch.ones(csr.E, 1, dtype=x.dtype, device=x.device)      # mutag.py:253
        for i in (1, 2, 3):
            g = lambda n: getattr(self, n % i)
            y = _gnnml1_block(x, csr, g('fc%d1'), g('conv%d1'), g('fc%d2'), g('fc%d3'), 2, 1)      # one launch (csrc/gml_gnnml1.hip)
            if y is None:
                y = torch.cat([F.relu(g('fc%d1')(x)), F.relu(g('conv%d1')(x, csr, ones)),
                               F.relu(g('fc%d2')(x)) * F.relu(g('fc%d3')(x))], 1)
            x = g('bn%d')(y)
        x = global_mean_pool(x, data)
        return tall_linear(F.relu(tall_linear(x, self.fc1)), self.fc2)


class GNNML1(torch.nn.Module):
    """GNNML1 as sr25.py:192-246 / graph8c.py:205-246 / mnist75.py:262-326 write it: three blocks of
         concat=False:  x <- act( fc_i1(x) + conv_i1(x) + fc_i2(x) * fc_i3(x) )            (the scripts' setting)
         concat=True :  x <- cat[ act(fc_i1 x), act(conv_i1 x), act(fc_i2 x * fc_i3 x) ]
    with conv_i1 = SpectConv(K=1, selfconn=False) over the RAW adjacency with unit edge values, then pooling and
    head 'lin10' (sr25 / graph8c: fc1: nin -> 10) or 'bn_mlp' (mnist75: bn1, relu(fc1: nin -> 32), log_softmax(fc2: 32 -> 10)).
    Same attribute names as the reference, so its state_dict loads.  (mutag.py's variant -- relu on the factors, BatchNorm
    per block -- is GNNML1Mutag.)"""

    def __init__(self, ninp, nout=64, concat=False, act='tanh', pool='add', head='lin10', nclass=10):
        super().__init__()
        self.concat, self.pool, self.head = concat, pool, head
        self.act = {'tanh': torch.tanh, 'relu': F.relu}[act]
        nin = 3 * nout if concat else nout
        for i, fin in enumerate([ninp, nin, nin], start=1):
            setattr(self, 'conv%d1' % i, SpectConv(fin, nout, selfconn=False))
            for j in (1, 2, 3):
                setattr(self, 'fc%d%d' % (i, j), torch.nn.Linear(fin, nout))
        if head == 'lin10':
            self.fc1 = torch.nn.Linear(nin, nclass)
        else:
            self.bn1 = torch.nn.BatchNorm1d(nin)
            self.fc1 = torch.nn.Linear(nin, 32)
            self.fc2 = torch.nn.Linear(32, nclass)

    def forward(self, data):
        x = data.x
        csr = data.csr('edge_index')
        ones = torch.ones(csr.E, 1, dtype=x.dtype, device=x.device)      # sr25.py:231
        actid = 0 if self.act is torch.tanh else 1
        for i in (1, 2, 3):
            g = lambda n: getattr(self, n % i)
            y = _gnnml1_block(x, csr, g('fc%d1'), g('conv%d1'), g('fc%d2'), g('fc%d3'), 1 if self.concat else 0, actid)
            if y is not None:
                x = y
                continue
            a, c, h = g('fc%d1')(x), g('conv%d1')(x, csr, ones), g('fc%d2')(x) * g('fc%d3')(x)
            x = torch.cat([self.act(a), self.act(c), self.act(h)], 1) if self.concat else self.act(a + c + h)
        x = {'add': global_add_pool, 'mean': global_mean_pool, 'max': global_max_pool}[self.pool](x, data)
        if self.head == 'lin10':
            return tall_linear(x, self.fc1)
        x = F.relu(tall_linear(self.bn1(x), self.fc1))
        return F.log_softmax(tall_linear(x, self.fc2), dim=1)


def sr25_gnnml1(ninp=2):                   # sr25.py:192-246 (nout = 64, sum form, tanh, add-pool, fc1 -> 10)
    return GNNML1(ninp, 64, concat=False, act='tanh', pool='add', head='lin10')


def mnist75_gnnml1(ninp=3):                # mnist75.py:262-326 (relu, mean-pool, bn1, 32 -> 10; dropout p = 0.1 omitted: eval)
    return GNNML1(ninp, 64, concat=False, act='relu', pool='mean', head='bn_mlp')


def zinc_gnnml3(ninp=25, ne=8):            # Zinc12k.py:316-329
    return GNNML3(ninp, ne, 30, 2, 4)


def counting_gnnml3(ninp=2, ne=12):        # counting.py:343-358
    return GNNML3(ninp, ne, 16, 16, 5)


def sr25_gnnml3(ninp=2, ne=6):             # sr25.py:252-262
    return GNNML3(ninp, ne, 32, 16, 3, head='tanh10')


def mnist_gnnml3(ninp=2, ne=6, dense_n=0):  # mnist75_gnnml3_tf.py:62, libs/models_tf.py:223-268 (DSGCNN)
    return GNNML3(ninp, ne, [64, 128, 128], 0, 3, learnedge=False, pool='mean', head='mlp32', nclass=10,
                  readout_bn=True, dense_n=dense_n)


def mutag_gnnml3(ninp=8, ne=4):            # mutag.py:272-288
    return GNNML3(ninp, ne, 24, 24, 3, learnedge=False, bn=True, pool='mean')


def zinc_loss(pre, y):                     # Zinc12k.py:365
    return F.l1_loss(pre, y.unsqueeze(-1), reduction='sum')


def zinc_step_loss(model, data, valid=None, loss_sum=None):
    """zinc_loss(model(data), data.y) -- with the head and the loss as ONE launch each way where the batch is small enough for it
    (functional.HeadL1Function: the reference's batch 64; a static batch's padding graph and absent slots are masked through
    `valid` / data.graph_valid).  Same value and gradients up to summation order; large batches take the general path."""
    from . import functional as Fn
    valid = valid if valid is not None else getattr(data, 'graph_valid', None)
    if model.head == 'mlp32' and model.fc2.weight.size(0) == 1 and data.x.is_cuda:
        nl = int(data.y.numel()) if valid is None else int(valid.numel())
        ng = int(data.num_graphs) if hasattr(data, 'num_graphs') else nl
        small = ng <= 256                                   # (the one-workgroup head: its rows beyond nl get a zero gradient)
        x = model.features(data, pad_grad_zero=small and nl <= ng)
        if Fn.head_l1_supported(x, model.fc1.weight, model.fc2.weight) and nl <= x.size(0):
            y = data.y[:nl].float().contiguous()
            return Fn.HeadL1Function.apply(x, y, valid, model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias, loss_sum)
        if small and nl <= ng:                              # (not the fused head after all: the promise above does not hold -- redo)
            x = model.features(data)
        if Fn.head_l1_big_supported(x, model.fc1.weight, model.fc2.weight) and nl <= x.size(0):
            # any batch size: one pass + a fold each way (rows beyond nl -- a static batch's padding graph -- get a zero gradient)
            y = data.y[:nl].float().contiguous()
            return Fn.HeadL1BigFunction.apply(x, y, valid, model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias, loss_sum)
        pre = tall_linear(F.relu(tall_linear(x, model.fc1)), model.fc2)
    else:
        pre = model(data)
    if valid is not None:
        nl = int(valid.numel())
        l = ((pre[:nl, 0] - data.y[:nl]).abs() * valid).sum()
    else:
        l = zinc_loss(pre, data.y)
    if loss_sum is not None:
        loss_sum.add_(l.detach())
    return l


def counting_loss(pre, y):                 # counting.py:411
    return torch.square(pre - y.view(-1, 1)).sum()


def mnist_loss(pre, y):                    # libs/metrics_tf.py softmax cross entropy, batch mean
    return F.cross_entropy(pre, y.long())


def mutag_loss(pre, y):                    # mutag.py:345-348
    return F.binary_cross_entropy(torch.sigmoid(pre)[:, 0], y, reduction='sum')
