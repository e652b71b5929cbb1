"""Oracle (CPU, torch fp32) for SpectConv / SpectConCatConv / ML3Layer.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates, op for op and in the
same accumulation order, what /root/reference/libs/spect_conv.py computes on a
CPU tensor, including the part that lives in the un-vendored dependency
pytorch_geometric==1.6.1 (``MessagePassing.propagate`` with aggr='add',
flow='source_to_target', node_dim=0):

    x_j  = x.index_select(0, edge_index[0])            # gather at the SOURCE
    msg  = norm.view(-1, 1) * x_j                      # spect_conv.py:98-99
    h    = zeros(N, F).scatter_add_(0, edge_index[1])  # sum at the TARGET

Everything is written as pure functions of explicit parameter tensors so that
tests can feed the very same parameters to the HIP path; the small nn.Module
wrappers at the bottom exist for the CPU-baseline timing and model oracles.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# the PyG-1.6.1 slice: one support column -> one aggregated feature matrix
# --------------------------------------------------------------------------
def propagate_add(x, edge_index, norm):
    """h[t] = sum_{e: dst(e)=t} norm[e] * x[src(e)]  (edge order = summation order).

    reference call sites: spect_conv.py:77,87,89,150.
    """
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(0, src)
    msg = norm.view(-1, 1) * x_j
    out = torch.zeros(x.size(0), x.size(1), dtype=x.dtype, device=x.device)
    idx = dst.view(-1, 1).expand_as(msg)
    return out.scatter_add_(0, idx, msg)


# --------------------------------------------------------------------------
# SpectConv.forward                                   spect_conv.py:64-96
# --------------------------------------------------------------------------
def spectconv_forward(x, edge_index, edge_attr, weight, bias=None, selfconn=False,
                      depthwise=False, dsweight=None):
    """weight [K', Fin, Fout]; K' = K(+1 if selfconn) or 1 when depthwise.
    dsweight [K(+1), Fin] only when depthwise (spect_conv.py:43-46)."""
    if not depthwise:
        nsup = weight.size(0)
        out = 0
        if selfconn:                                    # :72-74
            out = torch.matmul(x, weight[-1])
            nsup -= 1
        for i in range(nsup):                           # :76-80
            h = propagate_add(x, edge_index, edge_attr[:, i])
            out = out + torch.matmul(h, weight[i])
    else:
        nsup = dsweight.size(0)
        out = 0
        if selfconn:                                    # :83-85
            out = x * dsweight[-1]
            nsup -= 1
        out = out + (1 + dsweight[0:1, :]) * propagate_add(x, edge_index, edge_attr[:, 0])   # :87
        for i in range(1, nsup):                        # :88-89
            out = out + dsweight[i:i + 1, :] * propagate_add(x, edge_index, edge_attr[:, i])
        out = torch.matmul(out, weight[0])              # :91
    if bias is not None:                                # :93-94
        out = out + bias
    return out


# --------------------------------------------------------------------------
# SpectConCatConv.forward                             spect_conv.py:137-158
# --------------------------------------------------------------------------
def spectconcat_forward(x, edge_index, edge_attr, weight, bias=None, selfconn=True):
    pieces = []
    nsup = weight.size(0)
    if selfconn:
        pieces.append(torch.matmul(x, weight[-1]))
        nsup -= 1
    for i in range(nsup):
        h = propagate_add(x, edge_index, edge_attr[:, i])
        pieces.append(torch.matmul(h, weight[i]))
    out = torch.cat(pieces, 1)
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------
# ML3Layer.forward                                    spect_conv.py:204-212
# --------------------------------------------------------------------------
def edge_mlp_forward(edge_attr, w1, w2, w3, w4):
    """relu(fc1_4(cat[relu(fc1_1 ea), tanh(fc1_2 ea)*tanh(fc1_3 ea)])); all bias-free
    Linear layers, weights in torch.nn.Linear layout [out, in] (:190-194, :206-207)."""
    t = torch.cat([F.relu(F.linear(edge_attr, w1)),
                   torch.tanh(F.linear(edge_attr, w2)) * torch.tanh(F.linear(edge_attr, w3))], 1)
    return F.relu(F.linear(t, w4))


def ml3layer_forward(x, edge_index, edge_attr, p, learnedge, nout2, relu_mask=None):
    """p: dict with the reference state_dict keys (fc1_1.weight ... conv1.weight,
    conv1.bias, fc11.weight, fc11.bias, fc12.weight, fc12.bias).
    relu_mask (checker only, oracle/parity_at_size.py): evaluate the layer with a GIVEN activation pattern of its conv columns
    (a = z * mask instead of relu(z)) -- the pattern another evaluation of the same layer (the device's) produced."""
    if learnedge:
        edge_attr = edge_mlp_forward(edge_attr, p['fc1_1.weight'], p['fc1_2.weight'],
                                     p['fc1_3.weight'], p['fc1_4.weight'])
    a = spectconv_forward(x, edge_index, edge_attr, p['conv1.weight'], p.get('conv1.bias'), selfconn=False)
    a = F.relu(a) if relu_mask is None else a * relu_mask.to(a.dtype)
    if nout2 > 0:
        b = torch.tanh(F.linear(x, p['fc11.weight'], p['fc11.bias'])) * \
            torch.tanh(F.linear(x, p['fc12.weight'], p['fc12.bias']))
        return torch.cat([a, b], 1)
    return a


# --------------------------------------------------------------------------
# parameter init                                      spect_conv.py:13-20,58-62
# --------------------------------------------------------------------------
def glorot_(t):
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-bound, bound)
    return t


# --------------------------------------------------------------------------
# nn.Module wrappers (state_dict layout == reference)
# --------------------------------------------------------------------------
class OracleSpectConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, K=1, selfconn=True, depthwise=False, bias=True):
        super().__init__()
        assert K > 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.selfconn, self.depthwise = selfconn, depthwise
        if selfconn:
            K = K + 1
        if depthwise:
            self.DSweight = torch.nn.Parameter(torch.empty(K, in_channels))
            self.nsup = K
            K = 1
        self.weight = torch.nn.Parameter(torch.empty(K, in_channels, out_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.weight)
        if self.bias is not None:
            self.bias.data.zero_()
        if self.depthwise:
            self.DSweight.data.zero_()

    def forward(self, x, edge_index, edge_attr, edge_weight=None, batch=None, lambda_max=None):
        return spectconv_forward(x, edge_index, edge_attr, self.weight, self.bias, self.selfconn,
                                 self.depthwise, self.DSweight if self.depthwise else None)


class OracleSpectConCatConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, K, selfconn=True, bias=True):
        super().__init__()
        assert K > 0
        self.in_channels, self.out_channels, self.selfconn = in_channels, out_channels, selfconn
        if selfconn:
            K = K + 1
        self.weight = torch.nn.Parameter(torch.empty(K, in_channels, out_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(K * out_channels))
        else:
            self.register_parameter('bias', None)
        glorot_(self.weight)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, edge_index, edge_attr, edge_weight=None, batch=None, lambda_max=None):
        return spectconcat_forward(x, edge_index, edge_attr, self.weight, self.bias, self.selfconn)


class OracleML3Layer(torch.nn.Module):
    def __init__(self, learnedge, nedgeinput, nedgeoutput, ninp, nout1, nout2):
        super().__init__()
        self.learnedge, self.nout2 = learnedge, nout2
        if learnedge:
            self.fc1_1 = torch.nn.Linear(nedgeinput, 2 * nedgeinput, bias=False)
            self.fc1_2 = torch.nn.Linear(nedgeinput, 2 * nedgeinput, bias=False)
            self.fc1_3 = torch.nn.Linear(nedgeinput, 2 * nedgeinput, bias=False)
            self.fc1_4 = torch.nn.Linear(4 * nedgeinput, nedgeoutput, bias=False)
        else:
            nedgeoutput = nedgeinput
        self.conv1 = OracleSpectConv(ninp, nout1, nedgeoutput, selfconn=False)
        if nout2 > 0:
            self.fc11 = torch.nn.Linear(ninp, nout2)
            self.fc12 = torch.nn.Linear(ninp, nout2)

    def forward(self, x, edge_index, edge_attr):
        p = dict(self.named_parameters())
        return ml3layer_forward(x, edge_index, edge_attr, p, self.learnedge, self.nout2, getattr(self, '_relu_mask', None))
