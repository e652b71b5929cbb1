import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_matlang_amd import SpectralDesign, collate, models, synthetic, functional as Fn
from gnn_matlang_amd.graph import Batch
dev = torch.device('cuda:0')
raw = synthetic.make_graphs('zinc', 512, seed=1)
gs = SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(raw)
b0 = collate(gs * 64).to(dev)           # 32768 graphs
fields = {k: v for k, v in b0.__dict__.items() if not k.startswith('_')}
m = models.zinc_gnnml3().to(dev)
def step():
    b = Batch(**fields)
    m.zero_grad(set_to_none=True)
    l = models.zinc_step_loss(m, b)
    with Fn.deferred_folds(list(m.parameters())):
        l.backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_time_total > 20 and e.name.startswith('aten::'):
        st = [s for s in (e.stack or []) if 'gnn_matlang_amd' in s][:1]
        rows.append((e.device_time_total, e.name, st))
for r in sorted(rows, reverse=True)[:25]: print(round(r[0]), r[1], r[2])
