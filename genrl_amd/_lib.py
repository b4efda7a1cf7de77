"""ctypes binding of libgenrl_hip.so — the C-ABI declared in include/genrl_hip.h.

argtypes are derived from the header itself so the binding cannot drift from it.  The product
path has no CPU fallback: importing ops without the built library raises."""
import ctypes, os, re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'genrl_hip.h')
SO = os.environ.get('GENRL_HIP_SO', os.path.join(HERE, 'libgenrl_hip.so'))   # override: kernel experiments

_CT = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])}"""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(int|long)\s+(genrl_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        al = []
        for a in [x.strip() for x in args.split(',') if x.strip() and x.strip() != 'void']:
            if '*' in a:
                al.append((ctypes.c_void_p, a.split('*')[-1].strip()))
            else:
                t, n = a.rsplit(' ', 1)
                al.append((_CT[t.replace('const', '').strip()], n))
        out[name] = (_CT[ret], al)
    return out


class GenrlHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise GenrlHipError(f'{SO} is missing: build it with `python -m genrl_amd.build` '
                                '(there is no CPU fallback for the hot path)')
        # torch bundles its own libamdhip64; it must be resident BEFORE this library is dlopen'ed so
        # that both share ONE HIP runtime (two runtimes in a process -> "no ROCm-capable device").
        import torch  # noqa: F401
        L = ctypes.CDLL(SO)
        for name, (ret, args) in parse_header().items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = [t for t, _ in args]
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        detail = ''
        if code == 2:
            try:
                fn = _lib.genrl_last_error
                fn.restype = ctypes.c_char_p
                detail = ': ' + fn().decode()
            except Exception:
                pass
        raise GenrlHipError(f'{what} failed with status {code}{detail}')
