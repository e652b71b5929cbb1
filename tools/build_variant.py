#!/usr/bin/env python3
"""Build a variant of libgml_hip.so with extra compiler flags into _ab/lib_<name>.so (own object directory, the
in-tree library is left alone):   python tools/build_variant.py <name> [-DFLAG ...]
Used for A/B runs on one GPU box (tools/ab_bench.sh) and for the phase-timing build (-DGML_BWD2_TIMING)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get('GML_CSRC') or os.path.join(ROOT, 'gnn_matlang_amd', 'csrc')   # GML_CSRC: another source tree (e.g. a checkout of HEAD)
INCLUDE = os.path.join(ROOT, 'include')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(os.environ.get('GML_VARIANT_OBJ', '/tmp/gml_variant_obj'), name)   # outside the tree: objects never travel to the GPU box
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, '_ab'), exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', CSRC, '-I', INCLUDE, '-Wno-unused-result'] + extra
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hm = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith('.h'))
    stamp = os.path.join(objdir, 'FLAGS')
    same = os.path.exists(stamp) and open(stamp).read() == ' '.join(extra)

    def cc(src):
        obj = os.path.join(objdir, src[:-4] + '.o')
        sp = os.path.join(CSRC, src)
        if same and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hm):
            return obj
        r = subprocess.run([HIPCC] + flags + ['-c', sp, '-o', obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(src + '\n' + r.stderr)
        return obj
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, srcs))
    open(stamp, 'w').write(' '.join(extra))
    lib = os.path.join(ROOT, '_ab', 'lib_%s.so' % name)
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    print(lib)


if __name__ == '__main__':
    main()
