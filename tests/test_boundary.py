"""CPU checks of the drop-in boundary: the C-ABI library exports every symbol of
include/genrl_hip.h, the product modules reproduce the reference's weight contract, and the
product path fails loudly (no CPU fallback) when asked to compute without the MI355X."""
import ctypes, os
import pytest
import torch

import param_shapes
from oracle import genrl_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    from genrl_amd import _lib, build
    build.build(verbose=False)
    decl = _lib.parse_header()
    assert len(decl) >= 30
    L = ctypes.CDLL(_lib.SO)
    for name in decl:
        assert hasattr(L, name), name
    _lib.lib()   # argtypes bind


@pytest.mark.parametrize('tiny', [True, False])
def test_weight_contract(tiny):
    from genrl_amd import config
    over = config.tiny_overrides() if tiny else {}
    cfg = config.default_cfg(2, 16, device='cpu', **over)
    ag = config.make_agent(cfg)
    mine = {k: tuple(v.shape) for k, v in ag.state_dict().items()}
    ocfg = O.make_cfg(stoch=4, discrete=4, deter=32, hidden=32, units=32, cnn_depth=4) if tiny else O.make_cfg()
    ref = param_shapes.agent_param_shapes(ocfg)
    assert mine == ref
    assert not any(p.requires_grad for p in ag.parameters())
    if not tiny:
        n = lambda pre: sum(v.numel() for k, v in ag.state_dict().items() if k.startswith(pre))
        assert n('wm.') == 43328162 and n('_imag_behavior.actor.') == 5275668 and n('_imag_behavior.critic.') == 5516543


def test_no_cpu_fallback():
    from genrl_amd import ops, _lib
    with pytest.raises(_lib.GenrlHipError):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8), None)


def test_agent_is_picklable():
    import io
    from genrl_amd import config
    cfg = config.default_cfg(2, 16, device='cpu', **config.tiny_overrides())
    ag = config.make_agent(cfg)
    buf = io.BytesIO(); torch.save(ag, buf); buf.seek(0)
    ag2 = torch.load(buf, weights_only=False)
    assert set(ag2.state_dict()) == set(ag.state_dict())


def test_dreamer_agent_weight_contract():
    from genrl_amd import config
    cfg = config.dreamer_cfg(2, 18, device='cpu', **config.dreamer_tiny_overrides())
    ag = config.make_dreamer_agent(cfg)
    ocfg = O.make_cfg(stoch=4, discrete=4, act_dim=6, deter=32, hidden=32, units=32, cnn_depth=4,
                      single_obs_posterior=False, decoder_inputs='feat')
    assert {k: tuple(v.shape) for k, v in ag.state_dict().items()} == param_shapes.agent_param_shapes(ocfg, dreamer=True)
