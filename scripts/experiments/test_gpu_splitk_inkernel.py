"""The in-kernel split-K reduction's test as it stood in tests/test_gpu_splitk.py up to commit 3447b6f (see README.md here)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(256, 1024, 1024), (256, 1024, 3072), (256, 3072, 1024), (128, 1024, 2048), (200, 136, 5000), (96, 1728, 20000),
                                   (64, 4096, 1024), (320, 512, 4096), (1024, 1024, 8192)])
@pytest.mark.parametrize('layout', ['kk', 'kr', 'rk'])
def test_in_kernel_reduction_is_the_reduce_launch(shape, layout):
    """split-K products with the reduction inside the kernel (genrl_splitk_inkernel(1); opt-in) against partial tiles + reduce launch:
    torch.equal, with bias and with accumulation, while a second stream runs split-K products of its own; and the counters are back at
    zero -- the same launch again gives the same bits"""
    from genrl_amd import ops
    from genrl_amd._lib import lib
    M, N, K = shape
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g); B = torch.randn(N, K, device='cuda', generator=g)
    bias = torch.randn(N, device='cuda', generator=g); C0 = torch.randn(M, N, device='cuda', generator=g)
    At, Bt = A.t().contiguous(), B.t().contiguous()
    s2 = torch.cuda.Stream()

    def run(on):
        prev = lib().genrl_splitk_inkernel(int(on))
        try:
            outs = []
            for rep in range(3):
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    A2 = torch.ones(64, 4096, device='cuda'); B2 = torch.ones(256, 4096, device='cuda'); C2 = torch.empty(64, 256, device='cuda')
                    for _ in range(4):
                        ops.sgemm(A2, 4096, 1, B2, 4096, 1, C2, 256, None, 64, 256, 4096)
                C = C0.clone()
                a = (A, K, 1) if layout[0] == 'k' else (At, 1, M)
                b = (B, K, 1) if layout[1] == 'k' else (Bt, 1, N)
                ops.sgemm(*a, *b, C, N, bias, M, N, K)
                ops.sgemm(*a, *b, C, N, None, M, N, K, accumulate=True)
                torch.cuda.current_stream().wait_stream(s2)
                assert torch.equal(C2, torch.full_like(C2, 4096.0))
                outs.append(C)
            assert all(torch.equal(outs[0], o) for o in outs[1:])
            return outs[0]
        finally:
            lib().genrl_splitk_inkernel(prev)
    two, one = run(False), run(True)
    assert torch.equal(one, two)
    ref = 2 * (A.double() @ B.double().T) + bias.double()
    assert (one.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()

