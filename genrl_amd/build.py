"""Build libgenrl_hip.so (gfx950) in-tree with hipcc.  `python -m genrl_amd.build`."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ['gemm.hip', 'gemm_planes.hip', 'gemm_planes_tn.hip', 'scan_coop.hip', 'rowops.hip', 'dist.hip', 'conv.hip', 'optim.hip', 'stats.hip']
OUT = os.path.join(HERE, 'libgenrl_hip.so')


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, 'csrc', f) for f in SRC + ['common.h']]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-c++20-extensions', '-shared', '-fPIC',
           '-I', os.path.join(os.path.dirname(HERE), 'include'),
           '-o', OUT] + [os.path.join(HERE, 'csrc', f) for f in SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
