#!/usr/bin/env python3
"""Turn one tools/gpu_profile.sh pass (gpurun_out/<tag>_*) into the committed profiles/ artefacts.
   python tools/make_profiles.py <tag, e.g. r02h> <prefix in profiles/, e.g. r02_h> [note]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ''
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT).decode().strip()


def run(*a):
    return subprocess.check_output([sys.executable] + list(a), cwd=ROOT).decode()


for ext in ('json', 'log'):
    src = os.path.join(G, '%s_bench.%s' % (tag, ext))
    if os.path.exists(src):
        open(os.path.join(P, '%s_bench_default.%s' % (pre, ext)), 'w').write(open(src).read())
src = os.path.join(G, '%s_bench_extras.json' % tag)                 # the full record behind the compact line (round 6)
if os.path.exists(src):
    open(os.path.join(P, '%s_bench_extras.json' % pre), 'w').write(open(src).read())
run('tools/rocprof_summary.py', os.path.join(G, '%s_stats' % tag, '%s_stats_results.db' % tag),
    os.path.join(P, '%s_kernel_stats_b131072.md' % pre))
run('tools/hbm_traffic.py', os.path.join(G, '%s_fetch' % tag, '%s_fetch_counter_collection.csv' % tag),
    os.path.join(G, '%s_write' % tag, '%s_write_counter_collection.csv' % tag), os.path.join(P, '%s_hbm_traffic_b131072' % pre),
    'bench.py (default batch 131072: ZINC GNNML3 train step), commit %s' % commit)
tj = json.load(open(os.path.join(P, '%s_hbm_traffic_b131072.json' % pre)))
tj['commit'] = commit
for name in ('%s_hbm_traffic_b131072.json' % pre, 'hbm_traffic_b131072.json'):
    json.dump(tj, open(os.path.join(P, name), 'w'), indent=1)
pm = run('tools/pmc_summary.py', *[os.path.join(G, '%s_pmc%s' % (tag, k), '%s_pmc%s_counter_collection.csv' % (tag, k)) for k in 'ABC'],
         'spectconv', 'edge_chain', 'ml3_split')
open(os.path.join(P, '%s_pmc_sq_lds.md' % pre), 'w').write(
    '# SQ counters per launch, %s (commit %s%s)\n\n```\n%s```\n' % (tag, commit, ': ' + note if note else '',
                                                                '\n'.join(l[:400] for l in pm.splitlines()) + '\n'))
mn = [json.load(open(os.path.join(G, '%s_mnist_%d.json' % (tag, n)))) for n in (1024, 4096) if os.path.exists(os.path.join(G, '%s_mnist_%d.json' % (tag, n)))]
if mn:
    json.dump({'commit': commit, 'tool': 'tools/bench_mnist.py <graphs>', 'round1_dense_lib_graphs_per_s': 404000, 'runs': mn},
              open(os.path.join(P, '%s_mnist75_bench.json' % pre), 'w'), indent=1)
oc = os.path.join(G, '%s_other_configs.jsonl' % tag)
if os.path.exists(oc):
    open(os.path.join(P, '%s_other_configs_bench.jsonl' % pre), 'w').write(open(oc).read())
fz = [open(os.path.join(G, '%s_fuzz_%s.log' % (tag, s))).read().strip().splitlines()[-1] for s in ('ml3', 'conv', 'spectral')
      if os.path.exists(os.path.join(G, '%s_fuzz_%s.log' % (tag, s)))]
if fz:
    open(os.path.join(P, '%s_fuzz_parity.jsonl' % pre), 'w').write('\n'.join(fz) + '\n')
for what, title in (('spmm', 'tools/bench_spmm.py (32,768 ZINC-like graphs, S = 8, Fin = 32: gml_spmm_fwd -> the 8-wave SpMM)'),
                    ('sr25', 'tools/bench_sr25_sweep.py --S 48 --nodes 500000 (real sr25 graphs tiled, S = 48: gml_spmm_fwd_ex -> gml_k_spmm3)')):
    fcsv = os.path.join(G, '%s_%s_fetch' % (tag, what), '%s_%s_fetch_counter_collection.csv' % (tag, what))
    wcsv = os.path.join(G, '%s_%s_write' % (tag, what), '%s_%s_write_counter_collection.csv' % (tag, what))
    if os.path.exists(fcsv) and os.path.exists(wcsv):
        run('tools/hbm_traffic.py', fcsv, wcsv, os.path.join(P, '%s_%s_hbm_traffic' % (pre, 'spmm' if what == 'spmm' else 'sr25_spmm')),
            title + ', commit %s' % commit)
for name in ('spmm.json', 'sr25_sweep.jsonl'):
    src = os.path.join(G, '%s_%s' % (tag, name))
    if os.path.exists(src):
        open(os.path.join(P, '%s_%s' % (pre, name.replace('spmm.json', 'spmm_bench.json'))), 'w').write(open(src).read())
b = json.load(open(os.path.join(P, '%s_bench_default.json' % pre)))
print('commit', commit, 'value', b['value'], 'ms/step', b['ms_per_step'])
print(b['kernels_ms_per_step'])
print('roofline', b['roofline']['frac'], b['roofline']['avg_launch_ms'], b['roofline'].get('traffic'))
for k in ('value_exact_fp32', 'fresh_batch', 'ref_batch', 'epoch_bs64'):
    if k in b:
        print(k, {kk: vv for kk, vv in b[k].items() if kk in ('value', 'ms_per_step', 'index_build_ms_per_batch')})
for k, v in list(tj['kernels'].items())[:8]:
    print('%-50s %8.1f MB x %d' % (k[:50], v['hbm_bytes_per_launch'] / 1e6, v['launches']))
