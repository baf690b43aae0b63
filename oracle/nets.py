"""Oracle: generator / discriminator forward in plain PyTorch float32 on the CPU (autograd gives the
backward).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement of reference ``model.py`` taking a state_dict (same keys as the reference):
``generator_forward`` follows model.py:83-98 + cLN model.py:180-205; ``discriminator_forward`` follows
model.py:118-132 with torch.nn.utils.spectral_norm's compute_weight (one power iteration in train
mode, sigma = u^T W v).  Pinned against tests/golden/model.npz (made from the imported reference).
"""
import numpy as np
import torch
import torch.nn.functional as F

G_LAYERS = [(128, 256, 5), (256, 256, 7), (256, 256, 7), (256, 256, 7), (256, 256, 7), (256, 64, 5)]


def cln(x, gain, bias, eps=1e-8):
    # model.py:180-205 ; x [B, C, T]
    B, C, T = x.shape
    step_sum = x.sum(1)
    step_pow_sum = x.pow(2).sum(1)
    cum_sum = torch.cumsum(step_sum, dim=1)
    cum_pow_sum = torch.cumsum(step_pow_sum, dim=1)
    cnt = torch.arange(C, C * (T + 1), C, dtype=x.dtype).view(1, -1)
    cum_mean = cum_sum / cnt
    cum_var = (cum_pow_sum - 2 * cum_mean * cum_sum) / cnt + cum_mean.pow(2)
    cum_std = (cum_var + eps).sqrt()
    x = (x - cum_mean.unsqueeze(1)) / cum_std.unsqueeze(1)
    return x * gain + bias


def generator_forward(sd, x, y):
    inp = torch.cat((x, y), dim=2).transpose(1, 2).contiguous()
    h = inp
    for l, (cin, cout, k) in enumerate(G_LAYERS):
        w, b = sd['convolutions.%d.0.conv.weight' % l], sd['convolutions.%d.0.conv.bias' % l]
        h = F.conv1d(h, w, b, padding=k - 1)[:, :, :-(k - 1)].contiguous()      # ConvNorm + Chomp1d
        h = cln(h, sd['convolutions.%d.2.gain0' % l], sd['convolutions.%d.2.bias0' % l])
        h = F.leaky_relu(h, 0.3)
    o = h.transpose(1, 2).contiguous()
    o = F.leaky_relu(F.linear(o, sd['fc1.weight'], sd['fc1.bias']), 0.3)
    o = F.linear(o, sd['fc2.weight'], sd['fc2.bias'])
    return torch.exp(3.2 * torch.tanh(o))


def _sn_weight(sd, name, train, new_bufs):
    w = sd[name + '.weight_orig']
    u, v = sd[name + '.weight_u'], sd[name + '.weight_v']
    wm = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
        new_bufs[name + '.weight_u'] = u
        new_bufs[name + '.weight_v'] = v
    sigma = torch.dot(u.detach(), torch.mv(wm, v.detach()))
    return w / sigma


def discriminator_forward(sd, x, train=False):
    """x [B, Cin, 64, T] -> (score [B, nout], updated u/v buffers when train=True)."""
    nb = {}
    h = x
    for l in range(5):
        h = F.leaky_relu(F.conv2d(h, _sn_weight(sd, 'layers.%d' % l, train, nb), sd['layers.%d.bias' % l]), 0.3)
    h = h.mean(dim=(2, 3))
    h = F.leaky_relu(F.linear(h, _sn_weight(sd, 'fc1', train, nb), sd['fc1.bias']), 0.3)
    h = F.leaky_relu(F.linear(h, _sn_weight(sd, 'fc2', train, nb), sd['fc2.bias']), 0.3)
    h = torch.sigmoid(F.linear(h, _sn_weight(sd, 'fc3', train, nb), sd['fc3.bias']))
    return h, nb


def energy_norm(mask, clean_band, p_power=1 / 6, inv_p=6):
    """train_nele.py:133-140 per utterance."""
    cp = torch.pow(clean_band.detach(), inv_p)
    beta_2 = cp.sum(dim=(1, 2), keepdim=True) / (mask * cp).sum(dim=(1, 2), keepdim=True)
    enh = clean_band * torch.pow(mask, p_power) * beta_2 ** p_power
    return enh, beta_2.reshape(-1)


def d_inputs(enh, noise, ref):
    """train_nele.py:143-146: [B,T,64] x3 -> [B,3,64,T]."""
    t = lambda a: a.unsqueeze(1).transpose(2, 3).contiguous()
    return torch.cat((t(enh), t(noise), t(ref)), dim=1)


def adam_reference(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (numpy float32)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    return (p - (lr / bc1) * (m / denom)).astype(np.float32), m.astype(np.float32), v.astype(np.float32)
