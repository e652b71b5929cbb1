"""Build libgml_hip.so (gfx950) in-tree with hipcc.  No torch, no JIT cache: the .so sits next to
the package so it travels with the source tree and is what the process demonstrably loads."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(os.path.dirname(PKG), 'include')
OBJDIR = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libgml_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', CSRC, '-I', INCLUDE,
         '-Wno-unused-result'] + os.environ.get('GML_CXXFLAGS', '').split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(INCLUDE, 'gml.h'))
    return max(os.path.getmtime(h) for h in hs)


# Build-time guard: the chunked ring forward (gml_k_spectconv_fwd4) differed from itself about once per 1e7 issues while
# its aggregation loop contained `v_pk_fma_f32 ... op_sel:[0,1,0]` (the low product reading the HIGH half of src1; DESIGN s4.1c).  The
# kernel steers the compiler away from that form with a register copy -- which a compiler update could undo silently.  Every
# (re)compile of a guarded source therefore also emits its device assembly and the build FAILS if the form is back.
ISA_GUARD = {'gml_fwd4_fam_a.hip': 'gml_k_spectconv_fwd4', 'gml_fwd4_fam_b.hip': 'gml_k_spectconv_fwd4',
             'gml_fwd4_fam_c.hip': 'gml_k_spectconv_fwd4'}
BAD_FORM = re.compile(r'v_pk_fma_f32 .*op_sel:\[0,1,0\]')


def check_isa(src):
    """Device assembly of a guarded source; raises when a guarded kernel contains the operand-select form above."""
    spath = os.path.join(CSRC, src)
    asm = os.path.join(OBJDIR, src[:-4] + '.s')
    r = subprocess.run([HIPCC] + FLAGS + ['-S', '--cuda-device-only', spath, '-o', asm], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc -S failed on %s:\n%s' % (src, r.stderr))
    kernel, bad = None, {}
    for line in open(asm):
        if line.startswith('_Z') and line.rstrip().split(':')[0].find(ISA_GUARD[src]) >= 0 and ':' in line:
            kernel = line.split(':')[0]
        elif line.startswith('_Z'):
            kernel = None
        elif kernel and BAD_FORM.search(line):
            bad[kernel] = bad.get(kernel, 0) + 1
    os.remove(asm)
    if bad:
        raise RuntimeError('%s: v_pk_fma_f32 ... op_sel:[0,1,0] is back in %s (gfx950 loses low products in this form inside the '
                           'divergent aggregation loop, DESIGN s4.1c): %s' % (src, ISA_GUARD[src], bad))


def _compile(src):
    obj = os.path.join(OBJDIR, src[:-4] + '.o')
    spath = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime()):
        return obj, False
    if src in ISA_GUARD and not os.environ.get('GML_NO_ISA_GUARD'):
        check_isa(src)
    cmd = [HIPCC] + FLAGS + ['-c', spath, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed on %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return obj, True


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('[gml] %s (%d sources, %s)' % (LIB, len(srcs), 'rebuilt' if rebuilt else 'up to date'))
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
