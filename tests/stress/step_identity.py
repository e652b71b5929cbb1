"""One process = a few train steps of a config on a seeded batch; prints a hash of the logits and every gradient of every step.
tests/stress/run_step_identity.sh runs it in many FRESH processes and counts the distinct hashes (expected: one per config)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gnn_matlang_amd import SpectralDesign, collate, models, synthetic
dev = torch.device('cuda:0')
cfg = os.environ.get('CFG', 'zinc')
if cfg == 'zinc':
    pool = SpectralDesign(recfield=2, dv=2, nfreq=7).design_many(synthetic.make_graphs('zinc', 4096, seed=11))
    ctor, loss = models.zinc_gnnml3, models.zinc_loss
elif cfg == 'counting':
    pool = SpectralDesign(recfield=1, dv=1, nfreq=10, adddegree=True, laplacien=False, addadj=True).design_many(synthetic.make_graphs('counting', 1024, seed=11))
    ctor, loss = (lambda: models.counting_gnnml3(2, 12)), models.counting_loss
elif cfg == 'sr25':          # the 15 real strongly-regular graphs tiled: 13 entries per row -> edge chunks forward, two-launch backward
    from gnn_matlang_amd import readers
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    raw = readers.load_sr(os.path.join(root, 'tests', 'golden', 'raw', 'sr251256.g6')) * 40
    pool = SpectralDesign(recfield=1, dv=2, nfreq=5, adddegree=True).design_many(raw)
    ctor, loss = (lambda: models.sr25_gnnml3(2, 6)), (lambda pre, y: pre.square().sum())
elif cfg == 'mutag':         # 48-wide hidden layers + BatchNorm (first layer on the exact kernels)
    rng = np.random.default_rng(2)
    raw = []
    for x, ei, y in synthetic.make_graphs('zinc', 2048, seed=11):
        x7 = np.zeros((x.shape[0], 7), dtype=np.float32)
        x7[np.arange(x.shape[0]), rng.integers(7, size=x.shape[0])] = 1
        raw.append((x7, ei, np.float32(rng.integers(2))))
    pool = SpectralDesign(recfield=1, dv=4, nfreq=3, adddegree=True).design_many(raw)
    ctor, loss = (lambda: models.mutag_gnnml3(8, 4)), models.mutag_loss
elif cfg in ('mutag_gnnml1', 'sr25_gnnml1'):   # round 5: the fused GNNML1 block (csrc/gml_gnnml1.hip), mutag.py factor form / sr25.py sum form
    if cfg == 'mutag_gnnml1':
        rng = np.random.default_rng(2)
        raw = []
        for x, ei, y in synthetic.make_graphs('zinc', 2048, seed=11):
            x7 = np.zeros((x.shape[0], 7), dtype=np.float32)
            x7[np.arange(x.shape[0]), rng.integers(7, size=x.shape[0])] = 1
            raw.append((x7, ei, np.float32(rng.integers(2))))
        pool = SpectralDesign(recfield=1, dv=4, nfreq=3, adddegree=True).design_many(raw)
        ctor, loss = (lambda: models.GNNML1Mutag(8)), models.mutag_loss
    else:
        from gnn_matlang_amd import readers
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        raw = readers.load_sr(os.path.join(root, 'tests', 'golden', 'raw', 'sr251256.g6')) * 40
        pool = SpectralDesign(recfield=1, dv=2, nfreq=5, adddegree=True).design_many(raw)
        ctor, loss = (lambda: models.sr25_gnnml1(2)), (lambda pre, y: pre.square().sum())
data = collate(pool).to(dev)
data.y = torch.rand(data.num_graphs, generator=torch.Generator().manual_seed(3)).to(dev) if cfg in ('zinc', 'sr25', 'sr25_gnnml1') else data.y.to(dev)
torch.manual_seed(5)
m = ctor().to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
h = hashlib.sha256()
for step in range(int(os.environ.get('STEPS', 4))):
    opt.zero_grad(set_to_none=True)
    pre = m(data)
    l = loss(pre, data.y)
    l.backward()
    h.update(pre.detach().cpu().numpy().tobytes())
    for p in m.parameters():
        h.update(p.grad.detach().cpu().numpy().tobytes())
    opt.step()
print(cfg, 'nodes', int(data.x.size(0)), 'hash', h.hexdigest()[:16])
