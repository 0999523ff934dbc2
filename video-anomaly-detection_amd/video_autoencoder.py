"""Drop-in `VideoAutoencoder` (reference models/video_autoencoder.py:279-384) on the MI355X HIP path.

Module tree, constructor arguments and state_dict keys follow the reference
(`encoder.encoder.{0,1,4,5,8,9,12,13}`, `convlstm.cells.N.conv`, optional `proj`,
`decoder.decoder.{0,1,3,4,6,7,9}`).  Inference under `eval()` + `torch.no_grad()` runs only through
libvad_hip.so; `train()` / autograd keep the stock torch.nn composition the reference's
train_video.py differentiates through (reference train_video.py:44-65).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .autoencoder import LEAK, _HipScorer, _init_like_reference


class ConvLSTMCell(nn.Module):
    """One gate convolution over cat[x, h] -> (i, f, g, o) (reference models/video_autoencoder.py:24-91)."""

    def __init__(self, input_dim: int, hidden_dim: int, kernel_size: int = 3):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size=kernel_size,
                              padding=kernel_size // 2, bias=True)

    def forward(self, x, hidden_state):
        h_cur, c_cur = hidden_state
        i, f, g, o = torch.split(self.conv(torch.cat([x, h_cur], dim=1)), self.hidden_dim, dim=1)
        c_next = torch.sigmoid(f) * c_cur + torch.sigmoid(i) * torch.tanh(g)
        h_next = torch.sigmoid(o) * torch.tanh(c_next)
        return h_next, c_next

    def init_hidden(self, batch_size, height, width, device):
        shape = (batch_size, self.hidden_dim, height, width)
        return torch.zeros(shape, device=device), torch.zeros(shape, device=device)


class ConvLSTM(nn.Module):
    """Stacked cells; layers outer, time inner (reference models/video_autoencoder.py:94-179)."""

    def __init__(self, input_dim: int, hidden_dims, kernel_size: int = 3, num_layers: int = 2,
                 batch_first: bool = True, return_all_layers: bool = False):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dims = hidden_dims if isinstance(hidden_dims, list) else [hidden_dims] * num_layers
        self.num_layers = len(self.hidden_dims)
        self.batch_first = batch_first
        self.return_all_layers = return_all_layers
        dims = [input_dim] + self.hidden_dims
        self.cells = nn.ModuleList(ConvLSTMCell(dims[i], dims[i + 1], kernel_size) for i in range(self.num_layers))

    def forward(self, x, hidden_state=None):
        if not self.batch_first:
            x = x.permute(1, 0, 2, 3, 4)
        b, t, _, h, w = x.size()
        if hidden_state is None:
            hidden_state = self._init_hidden(b, h, w, x.device)
        outputs, finals = [], []
        cur = x
        for layer, cell in enumerate(self.cells):
            hs, cs = hidden_state[layer]
            steps = []
            for ti in range(t):
                hs, cs = cell(cur[:, ti], (hs, cs))
                steps.append(hs)
            cur = torch.stack(steps, dim=1)
            outputs.append(cur)
            finals.append((hs, cs))
        if self.return_all_layers:
            return outputs, finals
        return outputs[-1], finals[-1]

    def _init_hidden(self, batch_size, height, width, device):
        return [cell.init_hidden(batch_size, height, width, device) for cell in self.cells]


def _per_frame(seq_module: nn.Module, x):
    """Apply a frame-wise stack to [B,C,H,W] or [B,T,C,H,W] (reference models/video_autoencoder.py:217-231)."""
    if x.dim() != 5:
        return seq_module(x)
    b, t = x.shape[:2]
    y = seq_module(x.reshape(b * t, *x.shape[2:]))
    return y.view(b, t, *y.shape[1:])


class VideoEncoder(nn.Module):
    """4 x [conv3x3-BN-LeakyReLU-MaxPool2] per frame (reference models/video_autoencoder.py:182-231)."""

    def __init__(self, in_channels: int = 3, latent_dim: int = 128):
        super().__init__()
        widths = [in_channels, 32, 64, 128, latent_dim]
        layers = []
        for i in range(4):
            layers += [nn.Conv2d(widths[i], widths[i + 1], kernel_size=3, padding=1), nn.BatchNorm2d(widths[i + 1]),
                       nn.LeakyReLU(LEAK, inplace=True), nn.MaxPool2d(2, 2)]
        self.encoder = nn.Sequential(*layers)

    def forward(self, x):
        return _per_frame(self.encoder, x)


class VideoDecoder(nn.Module):
    """3 x [convT2x2s2-BN-ReLU] + convT-Tanh per frame (reference models/video_autoencoder.py:234-276)."""

    def __init__(self, out_channels: int = 3, latent_dim: int = 128):
        super().__init__()
        widths = [latent_dim, 128, 64, 32]
        layers = []
        for i in range(3):
            layers += [nn.ConvTranspose2d(widths[i], widths[i + 1], kernel_size=2, stride=2),
                       nn.BatchNorm2d(widths[i + 1]), nn.ReLU(inplace=True)]
        layers += [nn.ConvTranspose2d(widths[3], out_channels, kernel_size=2, stride=2), nn.Tanh()]
        self.decoder = nn.Sequential(*layers)

    def forward(self, x):
        return _per_frame(self.decoder, x)


class VideoAutoencoder(nn.Module):
    """Reference `VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128, lstm_num_layers=2)`
    (models/video_autoencoder.py:279-384)."""

    #: clips per launch group
    chunk = 64
    #: "fp32" (default), "split" or "winograd" (the encoder's 3x3 convolutions; the ConvLSTM cell stays direct): see ConvAutoencoder.precision
    precision = "fp32"

    def __init__(self, in_channels: int = 3, latent_dim: int = 128, lstm_hidden_dim: int = 128,
                 lstm_num_layers: int = 2):
        super().__init__()
        self.in_channels = in_channels
        self.latent_dim = latent_dim
        self.lstm_hidden_dim = lstm_hidden_dim
        self.lstm_num_layers = lstm_num_layers
        self.encoder = VideoEncoder(in_channels, latent_dim)
        self.convlstm = ConvLSTM(input_dim=latent_dim, hidden_dims=[lstm_hidden_dim] * lstm_num_layers,
                                 kernel_size=3, num_layers=lstm_num_layers, batch_first=True,
                                 return_all_layers=False)
        self.proj = (nn.Conv2d(lstm_hidden_dim, latent_dim, kernel_size=1)
                     if lstm_hidden_dim != latent_dim else nn.Identity())
        self.decoder = VideoDecoder(in_channels, latent_dim)
        _init_like_reference(self)
        self._hip = _HipScorer()

    # ------------------------------------------------------------------ HIP path
    def _use_hip(self) -> bool:
        if not self.training and torch.is_grad_enabled():
            _HipScorer.warn_eval_with_grad(self)
        return not self.training and not torch.is_grad_enabled()

    def _packed(self, device) -> torch.Tensor:
        l = hip.lib()
        mode = hip.precision_mode(self.precision)
        if mode in (hip.PREC_BF16, hip.PREC_BF16S):
            raise hip.VadError("precision 'bf16' / 'bf16_operands' / 'bf16_tensors' is a training mode (VideoTrainer / ImageTrainer): scoring is exact 'fp32', 'split' or 'winograd'")
        key = (mode,) + _HipScorer.state_key(self)
        if self._hip.key != key or self._hip.packed is None or self._hip.packed.device != device:
            kc = _HipScorer.kernel_channels(self.in_channels)
            n = (l.vad_vid_packed_floats_c(kc, self.latent_dim, self.lstm_hidden_dim, self.lstm_num_layers)
                 if 1 <= self.in_channels <= hip.MAX_IN_CHANNELS else 0)
            if n == 0:
                raise hip.VadError(
                    f"VideoAutoencoder(in_channels={self.in_channels}, latent_dim={self.latent_dim}, "
                    f"lstm_hidden_dim={self.lstm_hidden_dim}, lstm_num_layers={self.lstm_num_layers}) is not "
                    f"supported by the HIP path (needs 1 <= in_channels <= {hip.MAX_IN_CHANNELS}, latent_dim and lstm_hidden_dim in "
                    f"[1, {hip.MAX_WIDTH}], 1 <= layers <= 8)")
            params = _HipScorer.widen_to_rgb(_HipScorer.float_params(self), self.in_channels, last_transposed=True)
            blob = np.empty(n, dtype=np.float32)
            hip.check(l.vad_vid_pack_c(hip.pointer_array(params), len(params), kc, self.latent_dim,
                                       self.lstm_hidden_dim, self.lstm_num_layers, mode, blob.ctypes.data), "vad_vid_pack")
            self._hip.packed = torch.from_numpy(blob).to(device)
            self._hip.key = key
            self._hip.mode = mode
        return self._hip.packed

    def invalidate_packed(self) -> None:
        """Drop the packed-weight cache (see ConvAutoencoder.invalidate_packed)."""
        self._hip.key = None

    def train(self, mode: bool = True):
        if mode != self.training:
            self._hip.key = None
        return super().train(mode)

    def capture(self, x: torch.Tensor, seq=True, frame=True, errmap=False, recon=False) -> "hip.CapturedCall":
        """Capture ONE scoring call on clips shaped like `x` into a hipGraph and return the replayable call (see
        ConvAutoencoder.capture).  At the reference's sizes (4 clips x 16 frames, evaluate_video.py:416; one window,
        evaluate_video.py:344) a call is ~45 short launches on two streams (the ConvLSTM layer wavefront is captured with
        its fork / join); replaying them as one graph removes the per-launch host cost."""
        if not self._use_hip():
            raise hip.VadError("capture is an inference entry point: call under eval() and torch.no_grad()")
        want = dict(seq=seq, frame=frame, errmap=errmap, recon=recon)
        _HipScorer.check_input(x, 5, self.in_channels)
        # 1- / 2-channel models: widened once outside the capture, outputs at kernel shape (see ConvAutoencoder.capture)
        xs3 = _HipScorer.widen_input(x.contiguous() if x.dtype == torch.uint8 else x.contiguous().float(), self.in_channels,
                                     x.dtype == torch.uint8).clone()
        eager = self._run_hip(xs3, prewidened=True, **want)
        out = {k: torch.empty_like(v) for k, v in eager.items()}
        view = xs3 if self.in_channels >= 3 else xs3[:, :, :self.in_channels]
        return hip.CapturedCall(lambda: self._run_hip(xs3, out=out, prewidened=True, **want), view, out,
                                keep=(self._hip.packed, self._hip.ws, xs3),
                                post=None if self.in_channels >= 3 else self._narrow_outputs)

    def _run_hip(self, x: torch.Tensor, seq=False, frame=False, errmap=False, recon=False, out=None, prewidened=False):
        """`prewidened` (captured calls): `x` already has the kernels' 3 planes and the outputs stay at kernel shape."""
        u8 = x.dtype == torch.uint8       # raw decoded frames [B,T,H,W,3]: normalised inside the kernels (row f-3)
        kc = _HipScorer.kernel_channels(self.in_channels)
        cin = kc if prewidened else self.in_channels
        _HipScorer.check_input(x, 5, cin)
        if u8:
            b, t, h, w, _ = x.shape
            x = x.contiguous()
        else:
            b, t, _, h, w = x.shape
            x = x.contiguous().float()
        l = hip.lib()
        dev = x.device
        packed = self._packed(dev)
        x = _HipScorer.widen_input(x, cin, u8)
        chunk = max(1, min(int(self.chunk), b))
        dims = (self.latent_dim, self.lstm_hidden_dim, self.lstm_num_layers)
        nbytes = l.vad_vid_workspace_bytes_c(chunk, t, h, w, *dims, kc)
        if nbytes == 0:
            raise hip.VadError(f"unsupported frame size {h}x{w}: H and W must be multiples of 16")
        ws = self._hip.workspace(nbytes, dev)
        if out is None:                                   # (a captured call hands in its own output tensors)
            out = {}
            if seq:
                out["seq"] = torch.empty(b, dtype=torch.float32, device=dev)
            if frame:
                out["frame"] = torch.empty(b, t, dtype=torch.float32, device=dev)
            if errmap:
                out["errmap"] = torch.empty(b, t, 1, h, w, dtype=torch.float32, device=dev)
            if recon:
                out["recon"] = torch.empty(b, t, kc, h, w, dtype=torch.float32, device=dev)
        if b == 0:                                        # an empty batch gives empty outputs, as the reference's modules do
            return out
        with torch.cuda.device(dev):
            hip.check(l.vad_vid_score_c(x.data_ptr(), hip.X_U8_NHWC if u8 else hip.X_F32_NCHW, self._hip.mode, kc, b, t, h, w, *dims,
                                        packed.data_ptr(), ws.data_ptr(), ws.numel(),
                                      chunk, hip.ptr(out.get("seq")), hip.ptr(out.get("frame")),
                                      hip.ptr(out.get("errmap")), hip.ptr(out.get("recon")), hip.current_stream()),
                      "vad_vid_score")
        hip.calls["vid_score"] += 1
        return out if prewidened else self._narrow_outputs(out)

    def _narrow_outputs(self, out: dict) -> dict:
        """Undo the 3-plane view of a 1- / 2-channel model (`_HipScorer.widen_to_rgb`): the kernels averaged over 3
        planes of which 3 - in_channels are exactly zero."""
        cin = self.in_channels
        if cin >= 3:
            return out
        out = dict(out)
        for k in ("seq", "frame", "errmap"):
            if k in out:
                out[k] = out[k] * (3.0 / cin)
        if "recon" in out:
            out["recon"] = out["recon"][:, :, :cin].contiguous()
        return out

    # ------------------------------------------------------------------ reference API
    def _torch_forward(self, x):
        encoded = self.encoder(x)
        lstm_out, _ = self.convlstm(encoded)
        b, t, c, h, w = lstm_out.size()
        projected = self.proj(lstm_out.reshape(b * t, c, h, w)).view(b, t, -1, h, w)
        return self.decoder(projected)

    def forward(self, x):
        """[B,T,C,H,W] -> reconstruction of the same shape (reference models/video_autoencoder.py:329-354)."""
        if self._use_hip():
            return self._run_hip(x, recon=True)["recon"]
        return self._torch_forward(x)

    def get_reconstruction_error(self, x, per_frame: bool = False, per_pixel: bool = False):
        """[B] clip scores, [B,T] frame scores or [B,T,1,H,W] maps; per_pixel wins when both flags are
        set (reference models/video_autoencoder.py:356-384)."""
        if self._use_hip():
            if per_pixel:
                return self._run_hip(x, errmap=True)["errmap"]
            if per_frame:
                return self._run_hip(x, frame=True)["frame"]
            return self._run_hip(x, seq=True)["seq"]
        error = (x - self._torch_forward(x)) ** 2
        if per_pixel:
            return error.mean(dim=2, keepdim=True)
        if per_frame:
            return error.mean(dim=[2, 3, 4])
        return error.mean(dim=[1, 2, 3, 4])

    #: windows per launch group of `score_windows`
    window_chunk = 64

    def score_windows(self, frames: torch.Tensor, sequence_length: int = 16, stride: int = 1,
                      errmap: bool = False, recon: bool = False):
        """Dense sliding-window scoring of ONE video `frames [F,3,H,W]`: window k = frames[k*stride : k*stride+T].
        Returns {'seq': [NW], 'frame': [NW,T]} (+ 'errmap' [NW,T,1,H,W], 'recon' [NW,T,3,H,W] on request), identical
        to calling `get_reconstruction_error` on every window as its own clip — which is what the reference's
        `generate_video_output` does with batch size 1 and three forwards per window (evaluate_video.py:322-352) —
        but each frame goes through the encoder once instead of once per window that contains it."""
        if not self._use_hip():
            raise hip.VadError("score_windows is an inference entry point: call under eval() and torch.no_grad()")
        u8 = frames.dtype == torch.uint8
        cin = self.in_channels
        _HipScorer.check_input(frames, 4, cin, what="frames")
        if u8:
            f, h, w, _ = frames.shape
        else:
            f, _, h, w = frames.shape
        t = int(sequence_length)
        l = hip.lib()
        nw = l.vad_vid_num_windows(f, t, int(stride))
        if nw <= 0 or stride > t:
            raise hip.VadError(f"need F >= T and 0 < stride <= T (F={f}, T={t}, stride={stride})")
        frames = frames.contiguous() if u8 else frames.contiguous().float()
        dev = frames.device
        packed = self._packed(dev)
        frames = _HipScorer.widen_input(frames, cin, u8)
        chunk = max(1, min(int(self.window_chunk), nw))
        dims = (self.latent_dim, self.lstm_hidden_dim, self.lstm_num_layers)
        kc = _HipScorer.kernel_channels(cin)
        nbytes = l.vad_vid_windows_workspace_bytes_c(chunk, t, int(stride), h, w, *dims, kc)
        if nbytes == 0:
            raise hip.VadError(f"unsupported frame size {h}x{w}: H and W must be multiples of 16")
        ws = self._hip.workspace(nbytes, dev)
        out = {"seq": torch.empty(nw, dtype=torch.float32, device=dev),
               "frame": torch.empty(nw, t, dtype=torch.float32, device=dev)}
        if errmap:
            out["errmap"] = torch.empty(nw, t, 1, h, w, dtype=torch.float32, device=dev)
        if recon:
            out["recon"] = torch.empty(nw, t, kc, h, w, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            hip.check(l.vad_vid_score_windows_c(frames.data_ptr(), hip.X_U8_NHWC if u8 else hip.X_F32_NCHW, self._hip.mode, kc, f, t,
                                                int(stride), h, w, *dims, packed.data_ptr(),
                                              ws.data_ptr(), ws.numel(), chunk, out["seq"].data_ptr(),
                                              out["frame"].data_ptr(), hip.ptr(out.get("errmap")),
                                              hip.ptr(out.get("recon")), hip.current_stream()), "vad_vid_score_windows")
        hip.calls["vid_score"] += 1
        return self._narrow_outputs(out)

    def score_seq_and_frames(self, x):
        """One pass returning {'seq': [B], 'frame': [B,T]} and nothing else (the reference's clip loop runs two forwards
        for these, evaluate_video.py:143-149); unlike `score_all` no reconstruction or error map is written."""
        if not self._use_hip():
            raise hip.VadError("score_seq_and_frames is an inference entry point: call under eval() and torch.no_grad()")
        return self._run_hip(x, seq=True, frame=True)

    def score_all(self, x):
        """One pass returning recon, error maps, frame and clip scores (the reference's dense video mode
        runs three forwards per window for these: evaluate_video.py:350-352)."""
        if not self._use_hip():
            raise hip.VadError("score_all is an inference entry point: call under eval() and torch.no_grad()")
        return self._run_hip(x, seq=True, frame=True, errmap=True, recon=True)
