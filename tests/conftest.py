import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# The plane-operand path (genrl_amd/ops_planes.py) is the one the headline size runs (1024 rollout rows); by default it switches in
# from 192 rows up.  The parity fixtures are tiny, so the suite lowers that threshold to keep pinning that path against the
# reference's vectors; test_gpu_planes.py::test_small_rollouts_take_the_fp32_operand_path covers the default policy and the
# fp32-operand rollout (ops._Rollout).
os.environ.setdefault('GENRL_PLANES_MIN_ROWS', '0')
# likewise the weight gradients on plane operands (genrl_gemm_h2_tn, from 2048 rows up by default): every product with a whole
# number of 64-row stages takes it in the suite
os.environ.setdefault('GENRL_TN_MIN_ROWS', '64')
# ... and the convolution products on plane operands (genrl_amd/ops_conv_planes.py, from 4096 pixel rows up by default)
os.environ.setdefault('GENRL_PLANES_CONV_MIN_ROWS', '0')
os.environ.setdefault('GENRL_PLANES_KR_MIN_K', '0')

# scripts/ holds measurement one-offs that run GPU work at import: never collect them
collect_ignore_glob = ['../scripts/*']
