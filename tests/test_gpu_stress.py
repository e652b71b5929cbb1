"""Reduced fresh-process repeat of the chunked ring forward: the full sweep is tests/stress/run_fwd4.sh (1,536 launches
in 128 processes, run by hand); this keeps a small version of it in the regular GPU suite, because the failure it guards
against (profiles/r04_fwd4_nondeterminism.txt: a row's last edge missing about once per 200 launches, only in some fresh processes)
cannot be seen by a repeat inside ONE process.  The build-time guard on the instruction form is gnn_matlang_amd/_build.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('S,fin,deg', [(6, 48, 13), (6, 32, 13), (4, 48, 40)])
def test_chunked_forward_fresh_processes_agree_with_the_oracle(S, fin, deg):
    env = dict(os.environ, DS=str(S), DF=str(fin), DEG=str(deg), DN='700', REPS='8')
    for proc in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'stress', 'fwd4_repeat.py')], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        last = r.stdout.strip().splitlines()[-1]
        assert last.endswith('failing reps 0'), (proc, r.stdout[-2000:])
