"""CPU: module construction, state_dict compatibility with the reference (keys + shapes), error behaviour."""
import os

import numpy as np
import pytest
import torch

M = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'model.npz'))


def test_state_dict_keys_and_shapes_match_reference():
    from nele_gan_amd.model import Discriminator, Discriminator_Quality, Generator_Conv1D_cLN
    G, D, Q = Generator_Conv1D_cLN(), Discriminator(), Discriminator_Quality()
    assert list(G.state_dict().keys()) == [str(k) for k in M['g_keys']]
    assert [str(tuple(v.shape)) for v in G.state_dict().values()] == [str(s) for s in M['g_shapes']]
    assert list(D.state_dict().keys()) == [str(k) for k in M['d_keys']]
    assert [str(tuple(v.shape)) for v in D.state_dict().values()] == [str(s) for s in M['d_shapes']]
    assert list(Q.state_dict().keys()) == [str(k) for k in M['dq_keys']]
    assert sum(p.numel() for p in G.parameters()) == 2093120
    assert sum(p.numel() for p in D.parameters()) == 343491
    assert sum(p.numel() for p in Q.parameters()) == 343466


def test_checkpoint_dict_round_trip(tmp_path):
    from nele_gan_amd.model import Discriminator, Generator_Conv1D_cLN
    G, D = Generator_Conv1D_cLN(), Discriminator()
    p = tmp_path / 'chkpt_1.pt'
    torch.save({'enhance-model': G.state_dict(), 'intel-model': D.state_dict()}, p)      # train_nele.py:274-277
    G2 = Generator_Conv1D_cLN()
    G2.load_state_dict(torch.load(p)['enhance-model'])                                   # inference.py:71-72
    for a, b in zip(G.state_dict().values(), G2.state_dict().values()):
        assert torch.equal(a, b)


def test_no_cpu_fallback():
    from nele_gan_amd.model import Discriminator, Generator_Conv1D_cLN
    G = Generator_Conv1D_cLN()
    with pytest.raises(RuntimeError):
        G(torch.zeros(1, 30, 64), torch.zeros(1, 30, 64))
    with pytest.raises(ValueError):
        G(torch.zeros(1, 30, 63), torch.zeros(1, 30, 63))
    with pytest.raises(ValueError):
        Discriminator()(torch.zeros(1, 2, 64, 30))


def test_c_abi_parameter_layouts_follow_named_parameters():
    """nele_gen_param_layout / nele_disc_param_layout (csrc/netplan.hip): the flat buffers the library-built plans read are laid out in
    nn.Module.parameters() order of the reference modules (model.py:43-82, 101-116) - offsets and total counts equal the mirror's."""
    import ctypes
    from nele_gan_amd import _lib, model
    G = model.Generator_Conv1D_cLN()
    offs = (ctypes.c_longlong * 28)()
    assert _lib.lib.nele_gen_param_layout(offs, 28) == 28
    o = 0
    for k, (name, p) in enumerate(G.named_parameters()):
        assert offs[k] == o, name
        o += p.numel()
    assert _lib.lib.nele_gen_param_count() == o
    for cls, cin, nout in ((model.Discriminator, 3, 3), (model.Discriminator, 3, 2), (model.Discriminator_Quality, 2, 2)):
        D = cls(nout=nout)
        offs = (ctypes.c_longlong * 16)()
        assert _lib.lib.nele_disc_param_layout(cin, nout, offs, 16) == 16
        o = 0
        for k, (name, p) in enumerate(D.named_parameters()):
            assert offs[k] == o, name
            o += p.numel()
        assert _lib.lib.nele_disc_param_count(cin, nout) == o
    # workspace sizes are host-side arithmetic (no GPU): monotone in the batch, larger with a backward pass
    a = _lib.lib.nele_gen_workspace_bytes(8, 251, 1, 0)
    b = _lib.lib.nele_gen_workspace_bytes(8, 251, 1, 1)
    c = _lib.lib.nele_gen_workspace_bytes(16, 251, 1, 1)
    assert 0 < a < b < c
    assert _lib.lib.nele_disc_workspace_bytes(8, 251, 3, 1) > 0 and _lib.lib.nele_disc_workspace_bytes(8, 20, 3, 1) == -1


def test_clean_state_cache_bookkeeping():
    """metrics.CleanStateCache: slots of a batch -> runs of consecutive pool rows (what restore / store copy in one piece)."""
    from nele_gan_amd.metrics import CleanStateCache
    runs = CleanStateCache._runs([(0, 3), (0, 4), (0, 5), (1, 0), (0, 7), (0, 8)])
    assert runs == [(0, 0, 3, 3), (3, 1, 0, 1), (4, 0, 7, 2)]
    c = CleanStateCache(1 << 20)
    assert c.lookup('siib', ('L', 1), ['a']) is None and c.misses == 1
    assert c.stats()['bytes'] == 0
