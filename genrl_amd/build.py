"""Build libgenrl_hip.so (gfx950) in-tree with hipcc.  `python -m genrl_amd.build`.

Every translation unit is compiled to its own object (in parallel, rebuilt only when it or a header changed) and the objects are linked
into the one shared library the ctypes stub loads; there are no device calls across translation units, so no relocatable device code."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ['gemm.hip', 'gemm_planes.hip', 'gemm_planes_tn.hip', 'rowops.hip', 'dist.hip', 'conv.hip', 'optim.hip', 'stats.hip', 'seq.hip']
HDR = ['common.h']
OUT = os.path.join(HERE, 'libgenrl_hip.so')
OBJ = os.path.join(HERE, 'csrc', 'build')


def _src(f):
    return os.path.join(HERE, 'csrc', f)


def _obj(f):
    return os.path.join(OBJ, f.replace('.hip', '.o'))


def _hdr_time():
    inc = os.path.join(os.path.dirname(HERE), 'include', 'genrl_hip.h')
    return max(os.path.getmtime(p) for p in [_src(h) for h in HDR] + [inc])


def _stale(f):
    o = _obj(f)
    return (not os.path.exists(o)) or os.path.getmtime(o) < max(os.path.getmtime(_src(f)), _hdr_time())


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(_src(d)) > t for d in SRC) or _hdr_time() > t


def build(force=False, verbose=True, extra=()):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-c++20-extensions', '-Wno-unused-value', '-fPIC',
             '-I', os.path.join(os.path.dirname(HERE), 'include')] + list(extra)
    todo = [f for f in SRC if force or _stale(f)]

    def cc(f):
        cmd = [hipcc] + flags + ['-c', _src(f), '-o', _obj(f)]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return f, r.returncode, r.stdout

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as ex:
        res = list(ex.map(cc, todo))
    bad = [(f, out) for f, rc, out in res if rc != 0]
    for f, out in bad:
        sys.stderr.write('--- %s\n%s\n' % (f, out))
    if bad:
        raise RuntimeError('hipcc failed on: ' + ', '.join(f for f, _ in bad))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + [_obj(f) for f in SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
