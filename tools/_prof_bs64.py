import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import SpectralDesign, models, synthetic, functional as Fn
from gnn_matlang_amd.dataset import DeviceDataset
from gnn_matlang_amd.optim import OneLaunchAdam
dev = torch.device('cuda:0')
raw = synthetic.make_graphs('zinc', 2000, seed=4242)
dsd = DeviceDataset.from_graphs(SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(raw), dev)
dsd.y = dsd.y.float(); dsd.prepare()
bd = dsd.bounds(64)
ids = torch.arange(64, device=dev)
cm = models.zinc_gnnml3().to(dev)
co = OneLaunchAdam(cm.parameters(), lr=1e-3)
acc = torch.zeros((), device=dev); one_ = torch.ones((), device=dev)
def step():
    b = dsd.batch_assembled(ids, bd)
    co.zero_grad(set_to_none=True)
    l = models.zinc_step_loss(cm, b, loss_sum=acc)
    with Fn.deferred_folds(None):
        l.backward(one_)
    co.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    n = e.name
    if any(k in n for k in ('copy_', 'clone', 'Memcpy', 'aten::add', 'aten::mul', 'aten::fill', 'aten::zero', 'elementwise', 'aten::index', 'aten::select', 'aten::to')) and 'aten::' in n:
        st = [s for s in (e.stack or []) if 'gnn_matlang_amd' in s or 'bench' in s or 'tools' in s][:2]
        print(n, e.input_shapes, st)
