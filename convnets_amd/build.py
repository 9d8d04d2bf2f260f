"""Builds convnets_amd/libcnk.so (the C-ABI library of include/cnk.h) with hipcc for gfx950.

    python -m convnets_amd.build [--force]

Each .hip translation unit is compiled to an object in parallel and linked into one shared library
kept IN-TREE next to this file (git-ignored, but shipped to the GPU box with the repo snapshot).
No torch / pybind dependency: the boundary is plain C.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libcnk.so')
SOURCES = ['api.hip', 'conv_igemm.hip', 'conv_pp.hip', 'conv_gemm1x1.hip', 'bnfold.hip', 'conv_fold3.hip', 'conv_wgrad.hip', 'conv_wgrad3.hip', 'conv_direct.hip', 'conv_smallc.hip', 'conv_rowpatch.hip', 'bn.hip', 'misc.hip', 'head.hip', 'dwconv.hip', 'sk.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-ffp-contract=off']


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'wgrad_chain.h'), os.path.join(HERE, '..', 'include', 'cnk.h')]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])
    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr:
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
