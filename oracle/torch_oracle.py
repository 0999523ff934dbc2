"""torch.nn.functional restatement of the reference's scoring path on CPU tensors.
Test infrastructure and bench.py's cpu_baseline only; never imported by the product.

Each function cites the reference lines it follows.  `state` is a {key: tensor} state dict with the
reference's key names, so no reference class is needed to run it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-5  # nn.BatchNorm2d default


def _bn(x, st, p):
    return F.batch_norm(x, st[p + ".running_mean"], st[p + ".running_var"], st[p + ".weight"], st[p + ".bias"],
                        training=False, eps=EPS)


def _conv(x, st, p, pad=1):
    return F.conv2d(x, st[p + ".weight"], st[p + ".bias"], padding=pad)


def _convt(x, st, p):
    return F.conv_transpose2d(x, st[p + ".weight"], st[p + ".bias"], stride=2)


def img_encode(st, x):
    """Encoder.forward (reference models/autoencoder.py:38-86)."""
    for i in range(1, 5):
        p = f"encoder.enc{i}"
        x = F.leaky_relu(_bn(_conv(x, st, p + ".0"), st, p + ".1"), 0.2)
        x = F.leaky_relu(_bn(_conv(x, st, p + ".3"), st, p + ".4"), 0.2)
        x = F.max_pool2d(x, 2, 2)
    return x


def img_forward(st, x):
    """ConvAutoencoder.forward (reference models/autoencoder.py:181-193, decoder :103-146)."""
    z = img_encode(st, x)
    for i in range(1, 5):
        p = f"decoder.dec{i}"
        z = F.relu(_bn(_convt(z, st, p + ".0"), st, p + ".1"))
        if i < 4:
            z = F.relu(_bn(_conv(z, st, p + ".3"), st, p + ".4"))
        else:
            z = torch.tanh(_conv(z, st, p + ".3"))
    return z


def img_scores(st, x):
    """get_reconstruction_error, both modes (reference models/autoencoder.py:199-221)."""
    recon = img_forward(st, x)
    emap = ((x - recon) ** 2).mean(dim=1, keepdim=True)
    return {"recon": recon, "errmap": emap, "scores": emap.mean(dim=[1, 2, 3])}


def convlstm_cell(st, p, x, h, c):
    """ConvLSTMCell.forward (reference models/video_autoencoder.py:54-85)."""
    hid = h.shape[1]
    i, f, g, o = torch.split(_conv(torch.cat([x, h], dim=1), st, p + ".conv"), hid, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def vid_forward(st, x, hid: int, layers: int):
    """VideoAutoencoder.forward (reference models/video_autoencoder.py:329-354)."""
    b, t = x.shape[:2]
    z = x.reshape(b * t, *x.shape[2:])
    for idx in (0, 4, 8, 12):                                        # VideoEncoder (:191-215)
        z = F.max_pool2d(F.leaky_relu(_bn(_conv(z, st, f"encoder.encoder.{idx}"), st, f"encoder.encoder.{idx + 1}"), 0.2), 2, 2)
    z = z.view(b, t, *z.shape[1:])
    for l in range(layers):                                          # ConvLSTM (:144-166)
        h = torch.zeros(b, hid, *z.shape[3:])
        c = torch.zeros_like(h)
        outs = []
        for ti in range(t):
            h, c = convlstm_cell(st, f"convlstm.cells.{l}", z[:, ti], h, c)
            outs.append(h)
        z = torch.stack(outs, dim=1)
    z = z.reshape(b * t, *z.shape[2:])
    if "proj.weight" in st:                                          # (:311, :346-349)
        z = _conv(z, st, "proj", pad=0)
    for idx in (0, 3, 6):                                            # VideoDecoder (:242-256)
        z = F.relu(_bn(_convt(z, st, f"decoder.decoder.{idx}"), st, f"decoder.decoder.{idx + 1}"))
    z = torch.tanh(_convt(z, st, "decoder.decoder.9"))               # (:259-260)
    return z.view(b, t, *z.shape[1:])


def vid_scores(st, x, hid: int, layers: int):
    """get_reconstruction_error, all modes (reference models/video_autoencoder.py:356-384)."""
    recon = vid_forward(st, x, hid, layers)
    err = (x - recon) ** 2
    return {"recon": recon, "errmap": err.mean(dim=2, keepdim=True), "frame": err.mean(dim=[2, 3, 4]),
            "seq": err.mean(dim=[1, 2, 3, 4])}
