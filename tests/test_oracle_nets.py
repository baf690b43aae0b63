"""CPU: the torch-fp32 oracle of G / D against the goldens made from the imported reference modules."""
import os

import numpy as np
import pytest
import torch

from oracle import nets
from weights_recipe import digest, seeded_state_arrays

M = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'model.npz'))


def state(prefix, seed):
    keys = [str(k) for k in M[prefix + '_keys']]
    shapes = [eval(str(s)) for s in M[prefix + '_shapes']] if prefix + '_shapes' in M else None
    return keys, shapes, seed


def g_state():
    keys = [str(k) for k in M['g_keys']]
    shapes = [eval(str(s)) for s in M['g_shapes']]
    arrs = seeded_state_arrays(list(zip(keys, shapes)), 101)
    return {k: torch.from_numpy(v).requires_grad_(True) for k, v in arrs.items()}


def d_state(seed=202, cin=3, nout=3):
    keys = [str(k) for k in M['d_keys']]
    shapes = [eval(str(s)) for s in M['d_shapes']]
    if cin != 3 or nout != 3:
        shapes = [((s[0], cin) + s[2:] if k == 'layers.0.weight_orig' else s) for k, s in zip(keys, shapes)]
        shapes = [((cin,) if k == 'layers.0.weight_v' else s) for k, s in zip(keys, shapes)]
        shapes = [((nout,) + s[1:] if k in ('fc3.weight_orig',) else s) for k, s in zip(keys, shapes)]
        shapes = [((nout,) if k in ('fc3.bias', 'fc3.weight_u') else s) for k, s in zip(keys, shapes)]
    arrs = seeded_state_arrays(list(zip(keys, shapes)), seed)
    return {k: torch.from_numpy(v) for k, v in arrs.items()}


def check_grad(prefix, name, g):
    g = np.asarray(g)
    if prefix + name in M:
        np.testing.assert_allclose(g, M[prefix + name], rtol=2e-4, atol=1e-6 * np.abs(M[prefix + name]).max())
    else:
        s, a, smp = digest(g)
        assert s == pytest.approx(float(M[prefix + name + '#sum']), rel=1e-3, abs=1e-4 * float(M[prefix + name + '#abs']))
        assert a == pytest.approx(float(M[prefix + name + '#abs']), rel=1e-4)
        np.testing.assert_allclose(smp, M[prefix + name + '#smp'], rtol=5e-4, atol=1e-6 * np.abs(M[prefix + name + '#smp']).max())


def test_generator_forward_backward_matches_reference():
    torch.set_num_threads(2)
    sd = g_state()
    x = torch.from_numpy(M['x']).requires_grad_(True)
    y = torch.from_numpy(M['y']).requires_grad_(True)
    mask = nets.generator_forward(sd, x, y)
    np.testing.assert_allclose(mask.detach().numpy(), M['g_mask'], rtol=2e-5)
    (mask * torch.from_numpy(M['g_gw'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), M['g_dx'], rtol=2e-4, atol=1e-5 * np.abs(M['g_dx']).max())
    for k, v in sd.items():
        check_grad('g_grad.', k, v.grad.numpy())


def test_discriminator_eval_and_train_match_reference():
    sd = {k: v.clone().requires_grad_(not (k.endswith('_u') or k.endswith('_v'))) for k, v in d_state().items()}
    x = torch.from_numpy(M['d_in']).requires_grad_(True)
    sc, _ = nets.discriminator_forward(sd, x, train=False)
    np.testing.assert_allclose(sc.detach().numpy(), M['d_eval_score'], rtol=1e-5)
    loss = torch.nn.functional.mse_loss(sc, torch.from_numpy(M['d_tgt']))
    assert loss.item() == pytest.approx(float(M['d_eval_loss']), rel=1e-5)
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), M['d_eval_din_grad'], rtol=1e-3, atol=1e-5 * np.abs(M['d_eval_din_grad']).max())
    for k, v in sd.items():
        if v.requires_grad:
            check_grad('d_eval_grad.', k, v.grad.numpy())
    # train mode: one power iteration, buffers advance
    sd2 = d_state()
    sc2, nb = nets.discriminator_forward(sd2, torch.from_numpy(M['d_in']), train=True)
    np.testing.assert_allclose(sc2.numpy(), M['d_train_score'], rtol=1e-5)
    for k, v in nb.items():
        np.testing.assert_allclose(v.numpy(), M['d_train_buf.' + k], rtol=1e-5, atol=1e-7)


def test_discriminator_quality_eval():
    sd = d_state(303, cin=2, nout=2)
    sc, _ = nets.discriminator_forward(sd, torch.from_numpy(M['d_in'][:, [0, 2]].copy()), train=False)
    np.testing.assert_allclose(sc.numpy(), M['dq_eval_score'], rtol=1e-5)


def test_gstep_glue_matches_reference():
    sd = g_state()
    d3 = d_state()
    q3 = d_state(303, cin=2, nout=2)
    cb = torch.from_numpy(M['x'][:1])
    nb = torch.from_numpy(M['y'][:1])
    mask = nets.generator_forward(sd, cb, nb)
    enh, beta2 = nets.energy_norm(mask, cb)
    assert beta2.item() == pytest.approx(float(M['gstep_beta2']), rel=1e-5)
    np.testing.assert_allclose(enh.detach().numpy(), M['gstep_enh'], rtol=2e-5)
    di = nets.d_inputs(enh, nb, cb.detach())
    np.testing.assert_allclose(di.detach().numpy(), M['gstep_d_inputs'], rtol=2e-5)
    score, _ = nets.discriminator_forward(d3, di, train=True)
    score_q, _ = nets.discriminator_forward(q3, di[:, [0, 2]], train=True)
    np.testing.assert_allclose(score.detach().numpy(), M['gstep_score'], rtol=1e-5)
    np.testing.assert_allclose(score_q.detach().numpy(), M['gstep_score_q'], rtol=1e-5)
    mse = torch.nn.MSELoss()
    loss = mse(score, torch.ones(1, 3)) + 0.5 * mse(score_q, torch.ones(1, 2))
    assert loss.item() == pytest.approx(float(M['gstep_loss']), rel=1e-5)
    loss.backward()
    np.testing.assert_allclose(sd['fc2.weight'].grad.numpy(), M['gstep_grad_fc2_w'], rtol=2e-3,
                               atol=1e-5 * np.abs(M['gstep_grad_fc2_w']).max())
    np.testing.assert_allclose(sd['convolutions.5.0.conv.bias'].grad.numpy(), M['gstep_grad_c5_b'], rtol=2e-3,
                               atol=1e-5 * np.abs(M['gstep_grad_c5_b']).max())
