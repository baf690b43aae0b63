"""GPU: generator / discriminator HIP paths against the reference goldens and the torch-fp32 oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from weights_recipe import digest, seeded_state_arrays  # noqa: E402

M = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'model.npz'))


def load_recipe(module, seed):
    sd = module.state_dict()
    arrs = seeded_state_arrays([(k, tuple(v.shape)) for k, v in sd.items()], seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in arrs.items()})
    return module.cuda()


def check_grad(prefix, name, g, rtol=3e-4):
    g = g.detach().cpu().numpy()
    if prefix + name in M:
        ref = M[prefix + name]
        np.testing.assert_allclose(g, ref, rtol=rtol, atol=2e-5 * np.abs(ref).max())
    else:
        s, a, smp = digest(g)
        ra = float(M[prefix + name + '#abs'])
        assert s == pytest.approx(float(M[prefix + name + '#sum']), rel=1e-3, abs=2e-5 * ra)
        assert a == pytest.approx(ra, rel=2e-4)
        ref = M[prefix + name + '#smp']
        np.testing.assert_allclose(smp, ref, rtol=1e-3, atol=2e-5 * np.abs(ref).max())


@pytest.fixture(scope='module')
def mods():
    assert torch.cuda.is_available()
    from nele_gan_amd import model
    return model


def test_generator_matches_reference_golden(mods):
    G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
    x = torch.from_numpy(M['x']).cuda()
    y = torch.from_numpy(M['y']).cuda()
    mask = G(x, y)
    assert mask.shape == (2, 40, 64)
    np.testing.assert_allclose(mask.detach().cpu().numpy(), M['g_mask'], rtol=1e-4)   # f32 MFMA, reordered sums
    G.flat_parameters().grad.zero_()
    (mask * torch.from_numpy(M['g_gw']).cuda()).sum().backward()
    for k, p in G.named_parameters():
        check_grad('g_grad.', k, p.grad)


def test_generator_eval_no_grad_path(mods):
    G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
    G.eval()
    with torch.no_grad():
        mask = G(torch.from_numpy(M['x']).cuda(), torch.from_numpy(M['y']).cuda())
    assert not mask.requires_grad
    np.testing.assert_allclose(mask.cpu().numpy(), M['g_mask'], rtol=1e-4)


def test_discriminator_eval_matches_reference_golden(mods):
    D = load_recipe(mods.Discriminator(), 202)
    D.eval()
    x = torch.from_numpy(M['d_in']).cuda().requires_grad_(True)
    sc = D(x)
    np.testing.assert_allclose(sc.detach().cpu().numpy(), M['d_eval_score'], rtol=2e-5)
    loss = torch.nn.functional.mse_loss(sc, torch.from_numpy(M['d_tgt']).cuda())
    assert loss.item() == pytest.approx(float(M['d_eval_loss']), rel=2e-5)
    D.flat_parameters().grad.zero_()
    loss.backward()
    ref = M['d_eval_din_grad']
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max())
    for k, p in D.named_parameters():
        check_grad('d_eval_grad.', k, p.grad, rtol=2e-3)


def test_discriminator_train_power_iteration(mods):
    D = load_recipe(mods.Discriminator(), 202)
    D.train()
    x = torch.from_numpy(M['d_in']).cuda().requires_grad_(True)
    sc = D(x)
    np.testing.assert_allclose(sc.detach().cpu().numpy(), M['d_train_score'], rtol=2e-5)
    loss = torch.nn.functional.mse_loss(sc, torch.from_numpy(M['d_tgt']).cuda())
    D.flat_parameters().grad.zero_()
    loss.backward()
    for k, v in D.state_dict().items():
        if k.endswith('_u') or k.endswith('_v'):
            np.testing.assert_allclose(v.cpu().numpy(), M['d_train_buf.' + k], rtol=2e-5, atol=1e-7)
    for k, p in D.named_parameters():
        check_grad('d_train_grad.', k, p.grad, rtol=2e-3)
    ref = M['d_train_din_grad']
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max())


def test_discriminator_quality_eval(mods):
    Q = load_recipe(mods.Discriminator_Quality(), 303)
    Q.eval()
    with torch.no_grad():
        sc = Q(torch.from_numpy(M['d_in'][:, [0, 2]].copy()).cuda())
    np.testing.assert_allclose(sc.cpu().numpy(), M['dq_eval_score'], rtol=2e-5)


def test_gstep_glue_matches_reference_golden(mods):
    # train_nele.py:130-155 with the HIP energy-norm + pack kernel in place of the torch ops
    G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
    D = load_recipe(mods.Discriminator(), 202)
    Q = load_recipe(mods.Discriminator_Quality(), 303)
    cb = torch.from_numpy(M['x'][:1]).cuda()
    nb = torch.from_numpy(M['y'][:1]).cuda()
    G.flat_parameters().grad.zero_()
    mask = G(cb, nb)
    din, beta2 = mods.energy_norm_pack(mask, cb, nb)
    assert beta2.item() == pytest.approx(float(M['gstep_beta2']), rel=1e-4)
    d_in_ref = M['gstep_d_inputs']                                     # [1,3,64,T]
    got = din.detach().cpu().numpy()[0]                                # [64,T,4]
    np.testing.assert_allclose(np.transpose(got[:, :, :3], (2, 0, 1)), d_in_ref[0], rtol=1e-4)
    assert np.all(got[:, :, 3] == 0)
    score = D.forward_packed(din)
    # D_Qua input (enh, clean): channels 0 and 2 of the packed tensor
    din_q = torch.zeros_like(din)
    din_q[..., 0] = din[..., 0]
    din_q[..., 1] = din[..., 2]
    score_q = Q.forward_packed(din_q)
    np.testing.assert_allclose(score.detach().cpu().numpy(), M['gstep_score'], rtol=1e-4)
    np.testing.assert_allclose(score_q.detach().cpu().numpy(), M['gstep_score_q'], rtol=1e-4)
    mse = torch.nn.MSELoss()
    loss = mse(score, torch.ones(1, 3).cuda()) + 0.5 * mse(score_q, torch.ones(1, 2).cuda())
    assert loss.item() == pytest.approx(float(M['gstep_loss']), rel=1e-4)
    loss.backward()
    ref = M['gstep_grad_fc2_w']
    np.testing.assert_allclose(G.fc2.weight.grad.cpu().numpy(), ref, rtol=5e-3, atol=5e-5 * np.abs(ref).max())
    ref = M['gstep_grad_c5_b']
    np.testing.assert_allclose(G.convolutions[5][0].conv.bias.grad.cpu().numpy(), ref, rtol=5e-3, atol=5e-5 * np.abs(ref).max())
    g0 = G.convolutions[0][0].conv.weight.grad.double()
    assert g0.abs().sum().item() == pytest.approx(float(M['gstep_grad_c0_w_abs']), rel=2e-3)


@pytest.mark.parametrize('B,T', [(3, 57), (1, 132)])
def test_models_vs_oracle_other_shapes(mods, B, T):
    from oracle import nets
    rs = np.random.RandomState(B * 100 + T)
    G = load_recipe(mods.Generator_Conv1D_cLN(), 7)
    D = load_recipe(mods.Discriminator(nout=2), 8)
    x = (0.1 + 0.4 * rs.rand(B, T, 64)).astype(np.float32)
    y = (0.1 + 0.4 * rs.rand(B, T, 64)).astype(np.float32)
    tgt = rs.rand(B, 2).astype(np.float32)
    # oracle
    sdg = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in G.state_dict().items()}
    sdd = {k: v.detach().cpu().clone().requires_grad_(not (k.endswith('_u') or k.endswith('_v'))) for k, v in D.state_dict().items()}
    xo, yo = torch.from_numpy(x), torch.from_numpy(y)
    mo = nets.generator_forward(sdg, xo, yo)
    eo, b2o = nets.energy_norm(mo, xo)
    so, nbufs = nets.discriminator_forward(sdd, nets.d_inputs(eo, yo, xo), train=True)
    lo = torch.nn.functional.mse_loss(so, torch.from_numpy(tgt))
    lo.backward()
    # HIP
    D.train()
    G.flat_parameters().grad.zero_()
    D.flat_parameters().grad.zero_()
    xc, yc = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m = G(xc, yc)
    np.testing.assert_allclose(m.detach().cpu().numpy(), mo.detach().numpy(), rtol=2e-4)
    din, b2 = mods.energy_norm_pack(m, xc, yc)
    np.testing.assert_allclose(b2.cpu().numpy(), b2o.detach().numpy(), rtol=1e-4)
    s = D.forward_packed(din)
    np.testing.assert_allclose(s.detach().cpu().numpy(), so.detach().numpy(), rtol=1e-4)
    loss = torch.nn.functional.mse_loss(s, torch.from_numpy(tgt).cuda())
    loss.backward()
    for k, p in G.named_parameters():
        ref = sdg[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-3, atol=1e-4 * np.abs(ref).max(), err_msg='G ' + k)
    for k, p in D.named_parameters():
        ref = sdd[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-3, atol=1e-4 * np.abs(ref).max(), err_msg='D ' + k)


def test_frozen_discriminator_skips_weight_grads(mods):
    D = load_recipe(mods.Discriminator(), 202)
    D.weight_grad_enabled = False
    x = torch.from_numpy(M['d_in']).cuda().requires_grad_(True)
    D.flat_parameters().grad.zero_()
    D(x).sum().backward()
    assert float(D.flat_parameters().grad.abs().sum()) == 0.0
    assert float(x.grad.abs().sum()) > 0.0


def test_adam_matches_torch_formula(mods):
    from nele_gan_amd.optim import Adam
    from oracle.nets import adam_reference
    G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
    opt = Adam(G, lr=5e-4)
    fp = G.flat_parameters()
    rs = np.random.RandomState(0)
    p = fp.flat.cpu().numpy().copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for step in (1, 2, 3):
        g = (rs.randn(p.size) * 1e-2).astype(np.float32)
        fp.grad.copy_(torch.from_numpy(g))
        opt.step()
        p, m, v = adam_reference(p, g, m, v, 5e-4, step)
        np.testing.assert_allclose(fp.flat.cpu().numpy(), p, rtol=1e-6, atol=1e-7)
    opt.zero_grad()
    assert float(fp.grad.abs().sum()) == 0.0
    # parameters are views of the flat buffer: the module sees the update
    assert G.fc2.weight.data_ptr() >= fp.flat.data_ptr()


def test_discriminator_rejects_short_inputs(mods):
    D = mods.Discriminator().cuda()
    with pytest.raises(ValueError):
        D(torch.zeros(1, 3, 64, 20).cuda())


def test_bf16_operand_mode_close_to_f32(mods):
    # bf16 MFMA operands, float32 accumulation: documented tolerance 3e-2 relative (8-bit mantissa), scores 2e-3 absolute
    from nele_gan_amd import synth
    rs = np.random.RandomState(4)
    B, T = 2, 132
    din = torch.from_numpy((0.2 + 0.5 * rs.rand(B, 3, 64, T)).astype(np.float32)).cuda()
    tgt = torch.from_numpy(rs.rand(B, 3).astype(np.float32)).cuda()
    outs = {}
    for prec in ('f32', 'bf16'):
        D = load_recipe(mods.Discriminator(), 202)
        D.precision = prec
        D.train()
        x = din.clone().requires_grad_(True)
        D.flat_parameters().grad.zero_()
        sc = D(x)
        torch.nn.functional.mse_loss(sc, tgt).backward()
        outs[prec] = (sc.detach().cpu().numpy(), x.grad.cpu().numpy(), D.layers[4].weight_orig.grad.cpu().numpy().copy(),
                      D.layers[1].weight_orig.grad.cpu().numpy().copy())
    a, b = outs['f32'], outs['bf16']
    assert np.abs(a[0] - b[0]).max() < 2e-3
    # generator: bf16 MFMA operands in every Conv1d / Linear GEMM (forward, dgrad, wgrad)
    rs = np.random.RandomState(9)
    xg = torch.from_numpy((0.1 + 0.4 * rs.rand(2, 60, 64)).astype(np.float32)).cuda()
    yg = torch.from_numpy((0.1 + 0.4 * rs.rand(2, 60, 64)).astype(np.float32)).cuda()
    gw = torch.from_numpy(rs.randn(2, 60, 64).astype(np.float32)).cuda()
    gg = {}
    for prec in ('f32', 'bf16'):
        G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
        G.precision = prec
        G.flat_parameters().grad.zero_()
        mk = G(xg, yg)
        (mk * gw).sum().backward()
        gg[prec] = (mk.detach().cpu().numpy(), G.flat_parameters().grad.cpu().numpy().copy())
    np.testing.assert_allclose(gg['bf16'][0], gg['f32'][0], rtol=6e-2)            # mask = exp(3.2 tanh(.)): bf16 operand noise
    ga, gb = gg['f32'][1], gg['bf16'][1]                                           # bf16 through 6 conv layers + cLN: ~6 % L2
    assert np.linalg.norm(ga - gb) < 0.1 * np.linalg.norm(ga)
    assert (ga * gb).sum() > 0.995 * np.linalg.norm(ga) * np.linalg.norm(gb)
    for i in (1, 2, 3):
        scale = np.abs(a[i]).max()
        assert np.abs(a[i] - b[i]).max() < 3e-2 * scale, i


@pytest.mark.parametrize('T', [41, 77, 100, 133])
def test_bf16_tile_kernels_on_ragged_frame_counts(mods, T):
    """The 2-D tile conv / weight-gradient kernels on output widths that are not multiples of their 64-column tiles (and heights
    that are not multiples of 8): D forward + backward in bf16 against the float32 kernels, eval-mode spectral norm."""
    rs = np.random.RandomState(T)
    x = torch.from_numpy((0.2 + 0.5 * rs.rand(2, 3, 64, T)).astype(np.float32)).cuda()
    outs = {}
    for prec in ('f32', 'bf16'):
        D = load_recipe(mods.Discriminator(), 202).eval()
        D.precision = prec
        xi = x.clone().requires_grad_(True)
        D.flat_parameters().grad.zero_()
        sc = D(xi)
        (sc * torch.tensor([[1.0, -2.0, 0.5]], device='cuda')).sum().backward()
        outs[prec] = (sc.detach().cpu().numpy(), xi.grad.cpu().numpy(), D.flat_parameters().grad.cpu().numpy().copy())
    a, b = outs['f32'], outs['bf16']
    assert np.abs(a[0] - b[0]).max() < 3e-3
    for i in (1, 2):                                   # bf16 operand noise: a few % of the largest element, direction preserved
        assert np.isfinite(b[i]).all()
        assert np.abs(a[i] - b[i]).max() < 5e-2 * np.abs(a[i]).max(), i
        assert (a[i] * b[i]).sum() > 0.9999 * np.linalg.norm(a[i]) * np.linalg.norm(b[i]), i


def test_bf16_pooling_gradient_buffer_gives_the_float32_buffers_gradients(mods, monkeypatch):
    """bf16 mode stores the last conv layer's output gradient (the pooling gradient) as bfloat16 where the span / tile kernels take it
    (nele_gap_mlp_bwd_var16 -> nele_conv_span_bf16_a16 + nele_conv_wgrad_bf16_d16).  Both consumers round that operand to bf16 anyway, so
    everything the data-gradient chain produces - the input gradient and every other layer's weight gradient - must be BIT-identical to the
    float32-buffer path (ops.GRAD16 = False), conv5's weight gradient too (same bf16 operands, same accumulation order); only conv5's bias
    gradient sums bf16-rounded instead of float32 values - and since the pooling gradient takes just two values per (utterance, channel),
    that rounding does not average out: up to 2^-8 relative, inside the documented bf16-mode tolerance (3e-2)."""
    from nele_gan_amd import ops
    B, T = 3, 251
    torch.manual_seed(5)
    x = torch.rand(B, 3, 64, T, device='cuda') * 2
    res = {}
    monkeypatch.setattr(ops, 'CONV16', False)        # the round-2 kernels (float32 activations in memory): the fallback for geometries conv16 declines
    for flag in ('1', '0'):
        monkeypatch.setattr(ops, 'GRAD16', flag == '1')
        D = load_recipe(mods.Discriminator(), 33)
        D.precision = 'bf16'
        D.eval()
        xin = x.clone().requires_grad_(True)
        D(xin).pow(2).sum().backward()
        bf = next(iter(D._bufs.values()))
        assert bf.grad16_ok == (flag == '1') and (bf.gbuf16 is not None) == (flag == '1')
        res[flag] = (xin.grad.clone(), {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None})
    assert ops.grad16_supported.__doc__
    gin1, g1 = res['1']
    gin0, g0 = res['0']
    assert torch.equal(gin1, gin0)
    for k in g0:
        if k.endswith('layers.4.bias') or k.endswith('layers.4.conv.bias'):
            torch.testing.assert_close(g1[k], g0[k], rtol=4e-3, atol=4e-3 * float(g0[k].abs().max()))
        else:
            assert torch.equal(g1[k], g0[k]), k


def test_bf16_activations_in_memory_change_nothing_but_the_bias_gradients(mods, monkeypatch):
    """Round 3: in bf16 mode the activations of conv1..conv4 and the output gradients of conv2..conv5 are STORED as bfloat16
    (csrc/conv16.hip; nele_conv_wgrad_bf16_a16d16).  Every consumer of those tensors rounded them to bf16 while staging them, so the MFMA
    operands are the same numbers as with float32 buffers (ops.CONV16 = False: the round-2 kernels), accumulated over k in the same order:
    scores and input gradient must be BIT-identical, the weight gradients equal up to their summation order (round 4, see below).  Only the bias gradients of conv2..conv5 sum bf16-rounded
    instead of float32 output gradients (the same documented effect as for conv5 in round 2)."""
    from nele_gan_amd import ops
    B, T = 3, 251
    torch.manual_seed(6)
    x = torch.rand(B, 3, 64, T, device='cuda') * 2
    res = {}
    for flag in ('1', '0'):
        monkeypatch.setattr(ops, 'CONV16', flag == '1')
        D = load_recipe(mods.Discriminator(), 34)
        D.precision = 'bf16'
        D.eval()
        xin = x.clone().requires_grad_(True)
        score = D(xin)
        score.pow(2).sum().backward()
        bf = next(iter(D._bufs.values()))
        assert bf.c16 == (flag == '1')
        assert (bf.act[0].dtype == torch.bfloat16) == (flag == '1') and (bf.act[4].dtype == torch.bfloat16) == (flag == '1') and bf.gbuf[0].dtype == torch.float32
        res[flag] = (score.detach().clone(), xin.grad.clone(), {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None},
                     [a.float().clone() for a in bf.act], bf.pooled.clone())
    s1, gin1, g1, a1, p1 = res['1']
    s0, gin0, g0, a0, p0 = res['0']
    # the pooling fused into conv5's epilogue (per-wave float64 partial sums of the float32 results) against the separate pooling pass
    torch.testing.assert_close(p1, p0, rtol=1e-6, atol=1e-7)
    for l in range(5):                                      # activations: what is stored is the bf16 rounding of what the float32 path stores
        assert torch.equal(a1[l], a0[l].bfloat16().float()), 'activation of conv%d' % (l + 1)      # (conv5 too: pooled from the float32 results in the conv kernel)
    assert torch.equal(s1, s0) and torch.equal(gin1, gin0)
    for k in g0:
        if '.bias' in k and k.startswith('layers.') and not k.startswith('layers.0.'):
            torch.testing.assert_close(g1[k], g0[k], rtol=8e-3, atol=8e-3 * float(g0[k].abs().max()))
        elif 'weight' in k and k.startswith('layers.') and not k.startswith('layers.0.'):
            # round 4: with both operands bf16 in memory the weight gradients of conv2..conv5 run on conv_wgrad_dma_kernel (several kernel
            # rows per workgroup, other position-tile groups): the same bf16 products, float32 sums in another order
            torch.testing.assert_close(g1[k], g0[k], rtol=0, atol=4e-6 * float(g0[k].abs().max()))
        else:
            assert torch.equal(g1[k], g0[k]), k


@pytest.mark.parametrize('B,T', [(1, 21), (2, 37), (1, 52), (2, 53), (3, 85), (2, 149)])
def test_bf16_activation_kernels_at_edge_frame_counts(mods, B, T):
    """conv16 tiles are 4 (or 8) rows x 64 columns: 52 frames is the shortest utterance whose every layer runs on the bf16-in-memory kernels
    (the weight-gradient tile kernel wants >= 32 output columns: conv5's output is 44 x (T - 20)); 53 / 85 / 149 leave partial column tiles
    of every width class and partial row tiles in every layer.  Shorter utterances (T = 21 is the shortest D accepts) fall back, as a whole,
    to the float32-buffer kernels.  Against the float32 mode with the tolerances of bf16 operands (scores 2e-3; weight gradients 4e-2 and the
    input gradient - five layers of bf16 products deep - 1e-1 of the largest entry; measured 6e-2 at the BASELINE frame count on either
    bf16 path); bit-identity with the float32-buffer bf16 path holds where that path rounds every operand too (test above)."""
    torch.manual_seed(100 + T)
    x = torch.rand(B, 3, 64, T, device='cuda') * 2
    res = {}
    for prec in ('f32', 'bf16'):
        D = load_recipe(mods.Discriminator(), 35)
        D.precision = prec
        D.train()
        xin = x.clone().requires_grad_(True)
        score = D(xin)
        score.pow(2).sum().backward()
        if prec == 'bf16':
            assert next(iter(D._bufs.values())).c16 == (T >= 52)
        res[prec] = (score.detach().clone(), xin.grad.clone(), {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None})
    assert float((res['bf16'][0] - res['f32'][0]).abs().max()) <= 2e-3
    def close(a, b, tol, what):
        assert bool(torch.isfinite(a).all()), what
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-12, what
    close(res['bf16'][1], res['f32'][1], 1e-1, 'input gradient')
    for k, g0 in res['f32'][2].items():
        close(res['bf16'][2][k], g0, 4e-2, k)


# ------------------------------------------------------------------ round 5: fused generator layers on bf16 activations (csrc/glayer.hip)
def _torch_layer(a16, W, bias, gain, beta, K, slope=0.3):
    """One layer of model.py:83-91 in float64 on the bf16-rounded operands the kernel multiplies: causal Conv1d + cLN + LeakyReLU."""
    x = a16.double()                                                      # [B, T + K - 1, Cin], left-padded
    w = W.to(torch.bfloat16).double()                                     # [N, Cin, K]
    y = torch.nn.functional.conv1d(x.transpose(1, 2), w) + bias.double()[None, :, None]          # [B, N, T]
    B, N, T = y.shape
    s = y.sum(1).cumsum(1)
    q = (y * y).sum(1).cumsum(1)
    n = N * torch.arange(1, T + 1, device=y.device, dtype=torch.float64)[None]
    m = s / n
    var = (q - 2 * m * s) / n + m * m
    xh = (y - m[:, None]) / torch.sqrt(var[:, None] + 1e-8)
    o = xh * gain.double()[None, :, None] + beta.double()[None, :, None]
    o = torch.where(o > 0, o, slope * o)
    return y.transpose(1, 2), m, 1.0 / torch.sqrt(var + 1e-8), o.transpose(1, 2)


@pytest.mark.parametrize('B,T,cin,cout,K', [(2, 40, 128, 256, 5), (3, 251, 256, 256, 7), (2, 501, 256, 256, 7), (1, 700, 256, 64, 5), (2, 256, 64, 256, 5),
                                            (1, 257, 256, 256, 7)])
def test_fused_generator_layer_against_torch_float64(mods, B, T, cin, cout, K):
    """nele_glayer16_fwd (conv + bias + cLN + LeakyReLU in one launch, strips of 256 frames chained through the carry slots) against the
    same arithmetic in float64 on the same bf16-rounded operands: the raw convolution to float32 accumulation noise, mean / rstd / the
    activation to 1e-5, the bf16 output to one bf16 rounding.  T = 40 .. 700: one, two and three strips, ragged last strips."""
    from nele_gan_amd import ops
    from nele_gan_amd._lib import call, ptr, stream
    import ctypes
    assert ops.glayer16_supported(cin, cout, K)
    g = torch.Generator().manual_seed(B * 1000 + T)
    W = (torch.randn(cout, cin, K, generator=g) / np.sqrt(cin * K)).cuda()
    bias, gain, beta = (0.1 * torch.randn(cout, generator=g)).cuda(), (1 + 0.2 * torch.randn(cout, generator=g)).cuda(), (0.1 * torch.randn(cout, generator=g)).cuda()
    a = torch.randn(B, T, cin, generator=g).cuda()
    a16 = torch.zeros(B, T + K - 1, cin, dtype=torch.bfloat16, device='cuda')
    a16[:, K - 1:] = a.to(torch.bfloat16)
    # GEMM layout [N][K * Cin] in k order (tap, channel), then the fragment stream
    wg = W.permute(0, 2, 1).reshape(cout, K * cin).contiguous()
    wfr = torch.zeros(ops.glayer16_wfrag_elems(cin, cout, K), dtype=torch.bfloat16, device='cuda')
    call('nele_glayer16_weight_prep_batch', (ctypes.c_void_p * 2)(wg.data_ptr(), wfr.data_ptr()), (ctypes.c_int * 3)(cout, cin, K), 1, stream())
    padn = 6
    Y = torch.empty(B, T, cout, device='cuda'); mean = torch.empty(B, T, device='cuda'); rstd = torch.empty(B, T, device='cuda')
    out16 = torch.zeros(B, T + padn, cout, dtype=torch.bfloat16, device='cuda'); out32 = torch.zeros(B, T + padn, cout, device='cuda')
    carry = torch.zeros(max(32, int(ops._lib.lib.nele_glayer16_carry_bytes(B, T))), dtype=torch.uint8, device='cuda')
    for token in (1, 2):                                                  # twice on the same carry buffer: stale slots must not match
        call('nele_glayer16_fwd', ptr(a16), ptr(wfr), ptr(bias), ptr(gain), ptr(beta), ptr(Y), ptr(mean), ptr(rstd), ptr(out16), ptr(out32), ptr(carry),
             token, B, T, cin, cout, K, padn, 0.3, stream())
    ry, rm, rr, ro = _torch_layer(a16, W, bias, gain, beta, K)
    scale = float(ry.abs().max())
    assert float((Y.double() - ry).abs().max()) <= 2e-5 * scale           # float32 accumulation over up to 1792 products
    np.testing.assert_allclose(mean.cpu().numpy(), rm.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(rstd.cpu().numpy(), rr.cpu().numpy(), rtol=2e-5)
    assert float((out32[:, padn:].double() - ro).abs().max()) <= 1e-4 * float(ro.abs().max())
    assert not out32[:, :padn].any() and not out16[:, :padn].any()       # the padding rows are left alone
    assert torch.equal(out16[:, padn:], out32[:, padn:].to(torch.bfloat16))
    # outputs that are not asked for are not needed: evaluation writes the bf16 activation only
    o2 = torch.zeros_like(out16)
    call('nele_glayer16_fwd', ptr(a16), ptr(wfr), ptr(bias), ptr(gain), ptr(beta), None, None, None, ptr(o2), None, ptr(carry), 3, B, T, cin, cout, K, padn, 0.3,
         stream())
    assert torch.equal(o2, out16)
    # the convolution alone (the data-gradient form)
    d = torch.empty(B, T, cout, device='cuda')
    call('nele_glayer16_conv', ptr(a16), ptr(wfr), ptr(d), B, T, cin, cout, K, stream())
    assert float((d.double() - (ry - bias.double())).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize('B,T', [(1, 30), (2, 40), (3, 251), (2, 501), (1, 600)])
def test_fused_generator_equals_per_layer_kernels_in_bf16_mode(mods, B, T):
    """Generator_Conv1D_cLN in bf16 mode: the fused layer path (bf16 activations in memory) against the per-layer kernels (float32
    activations rounded while staged).  The same bf16 products; the float32 sums run in another order, so a layer's activations differ
    in the last float32 bits and a few of them round to the neighbouring bf16 value (2^-8 relative) on their way into the next layer:
    the masks (exp(3.2 tanh(.)) of the sixth layer's output) agree to a fraction of the bf16-against-float32 tolerance (6e-2, test above).
    A difference d between two float32 values flips their bf16 rounding with probability d / 2^-8, so a relative difference d becomes
    sqrt(d 2^-8) one layer on: measured 2e-7, 4e-6, 8e-5, 3e-4, 7e-4, 1.3e-3 over the six layers (the bf16-against-float32 difference
    of the same tensors: 2e-3 .. 6e-3).  The backward pass consumes the float32 copies the fused forward pass writes and amplifies the
    forward difference as it amplifies bf16's (LeakyReLU kinks, tanh'): gradients 3e-2 apart where bf16 and float32 are 8e-2 apart."""
    g = torch.Generator().manual_seed(7)
    x, y = torch.rand(B, T, 64, generator=g).cuda(), torch.rand(B, T, 64, generator=g).cuda()
    gw = torch.randn(B, T, 64, generator=g).cuda()
    res = {}
    for fused in (False, True):
        G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
        G.precision = 'bf16'
        G.fused = fused
        mask = G(x, y)
        G.flat_parameters().grad.zero_()
        (mask * gw).sum().backward()
        G.eval()
        with torch.no_grad():
            me = G(x, y)
        res[fused] = (mask.detach().clone(), G.flat_parameters().grad.clone(), me.clone())
    a, b = res[False], res[True]
    assert float((a[0] - b[0]).abs().max()) <= 1e-2 * float(a[0].abs().max())
    assert float((a[0] - b[0]).norm()) <= 1e-2 * float(a[0].norm())
    assert float((a[2] - b[2]).abs().max()) <= 1e-2 * float(a[2].abs().max())
    if T >= 32:
        assert torch.equal(b[0], b[2])                                    # train- and eval-mode forward passes of the fused path: the same kernels
    else:                                                                 # below 32 frames a training pass keeps the per-layer kernels (model._fused_for)
        assert torch.equal(b[0], a[0]) and float((b[0] - b[2]).abs().max()) <= 1e-2 * float(b[0].abs().max())
    assert float((a[1] - b[1]).norm()) <= 6e-2 * float(a[1].norm())


# ------------------------------------------------------------------ round 5: composite entry points (recorded job tables, csrc/plan.hip)
@pytest.mark.parametrize('prec', ['f32', 'bf16'])
def test_recorded_plans_replay_bit_identically(mods, prec):
    """nele_gen_fwd / nele_gen_bwd / nele_disc_fwd / nele_disc_bwd enqueue the job table that the host mirror recorded from its own per-layer
    loop the first time a shape came by.  Call-by-call (_lib.PLANS = False), the recording pass and every replay must give the same bits:
    masks, scores, input gradients and the flat parameter gradients of G and D, over two different batches (the per-call pointers - inputs,
    outputs, carry tokens - are the plan's slots), with D's weight gradients on (D-step) and off (G-step) and a padded batch."""
    from nele_gan_amd import _lib
    B, T = 3, 300                                                          # two strips of the fused generator layers
    g = torch.Generator().manual_seed(11)
    batches = [(torch.rand(B, T, 64, generator=g).cuda(), torch.rand(B, T, 64, generator=g).cuda(), torch.randn(B, T, 64, generator=g).cuda(),
                torch.rand(B, 64, T, 4, generator=g).cuda(), torch.randn(B, 3, generator=g).cuda()) for _ in range(3)]
    frames = torch.tensor([300, 180, 251], dtype=torch.int32, device='cuda')

    def run(plans):
        _lib.PLANS = plans
        try:
            G = load_recipe(mods.Generator_Conv1D_cLN(), 101)
            D = load_recipe(mods.Discriminator(), 202)
            G.precision = D.precision = prec
            out = []
            for it, (x, y, gw, din, ds) in enumerate(batches):
                mask = G(x, y)
                G.flat_parameters().grad.zero_()
                (mask * gw).sum().backward()
                d_in = din.clone().requires_grad_(True)
                D.weight_grad_enabled = it != 1                            # the G-step form in between
                D.flat_parameters().grad.zero_()
                score = D.forward_packed(d_in, frames if it == 2 else None)
                (score * ds).sum().backward()
                D.weight_grad_enabled = True
                out.append([t.detach().clone() for t in (mask, G.flat_parameters().grad, score, d_in.grad, D.flat_parameters().grad)])
            with torch.no_grad():
                G.eval()
                out.append([G(batches[0][0], batches[0][1]).clone(), G(batches[1][0], batches[1][1]).clone()])
            nplans = sum(len(b.plans) for b in G._bufs.values()) + sum(len(b.plans) for b in D._bufs.values())
            return out, nplans
        finally:
            _lib.PLANS = True

    live, n0 = run(False)
    rec, n1 = run(True)
    assert n0 == 0 and n1 >= 6                                             # G fwd (train, eval) + bwd, D fwd (2 forms) + bwd (3 forms)
    for a, b in zip(live, rec):
        for u, v in zip(a, b):
            assert torch.equal(u, v)


def test_plan_run_reports_errors_and_slot_mismatches(mods):
    from nele_gan_amd import _lib
    import ctypes
    j = _lib.nele_plan_job()
    j.op, j.nargs, j.stream = _lib.lib.nele_plan_op_id(b'nele_vec_add'), 4, 0
    for i in range(4):
        j.slot[i] = -1
    j.slot[0], j.slot[1] = 0, 1
    j.ival[2] = 1000
    h = ctypes.c_void_p()
    arr = (_lib.nele_plan_job * 1)(j)
    _lib.check(_lib.lib.nele_plan_create(arr, 1, 2, 1, ctypes.byref(h)), 'create')
    a, b = torch.ones(1000, device='cuda'), torch.full((1000,), 2.0, device='cuda')
    streams = (ctypes.c_void_p * 1)(_lib.stream())
    slots = (ctypes.c_longlong * 2)(a.data_ptr(), b.data_ptr())
    _lib.check(_lib.lib.nele_plan_run(h, streams, 1, slots, 2), 'run')
    assert float(a.sum()) == 3000.0
    assert _lib.lib.nele_plan_run(h, streams, 1, slots, 1) == -1           # too few slots
    slots[1] = 0
    assert _lib.lib.nele_plan_run(h, streams, 1, slots, 2) == -1           # the entry point's own argument check (null source)
    assert b'nele_vec_add' in _lib.lib.nele_last_error_string()
    j.nargs = 3
    h2 = ctypes.c_void_p()
    assert _lib.lib.nele_plan_create((_lib.nele_plan_job * 1)(j), 1, 2, 1, ctypes.byref(h2)) == -1
    _lib.check(_lib.lib.nele_plan_destroy(h), 'destroy')
