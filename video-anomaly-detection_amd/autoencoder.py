"""Drop-in `ConvAutoencoder` (reference models/autoencoder.py:149-221) on the MI355X HIP path.

Same constructor, `forward` / `get_latent` / `get_reconstruction_error` signatures, module tree and
state_dict keys as the reference, so a reference checkpoint strict-loads.  Parameters stay ordinary
`nn.Parameter`s in PyTorch layouts; the kernels read a derived blob (BatchNorm folded, MFMA operand
order) that is rebuilt whenever a parameter changes.

Inference (`eval()` under `torch.no_grad()`) runs ONLY through libvad_hip.so and raises if the
library or a GPU tensor is missing.  `train()` mode / autograd keep the stock torch.nn composition,
which is what the reference's train.py differentiates through (reference train.py:41-46); that is
outside the scoring hot path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import hip

LEAK = 0.2  # reference models/autoencoder.py:41


def _conv_bn_act(cin: int, cout: int, act: nn.Module) -> list:
    return [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), act]


class Encoder(nn.Module):
    """4 x [conv3x3-BN-LeakyReLU, conv3x3-BN-LeakyReLU, MaxPool2] (reference models/autoencoder.py:24-86).
    Sub-module names enc1..enc4 with Sequential indices 0,1,3,4 carrying state, as in the reference."""

    def __init__(self, in_channels: int = 3, latent_dim: int = 256):
        super().__init__()
        widths = [in_channels, 32, 64, 128, latent_dim]
        for i in range(4):
            layers = (_conv_bn_act(widths[i], widths[i + 1], nn.LeakyReLU(LEAK, inplace=True))
                      + _conv_bn_act(widths[i + 1], widths[i + 1], nn.LeakyReLU(LEAK, inplace=True))
                      + [nn.MaxPool2d(2, 2)])
            setattr(self, f"enc{i + 1}", nn.Sequential(*layers))

    def forward(self, x):
        for i in range(1, 5):
            x = getattr(self, f"enc{i}")(x)
        return x


class Decoder(nn.Module):
    """3 x [convT2x2s2-BN-ReLU, conv3x3-BN-ReLU] + [convT-BN-ReLU, conv3x3, Tanh]
    (reference models/autoencoder.py:89-146)."""

    def __init__(self, out_channels: int = 3, latent_dim: int = 256):
        super().__init__()
        widths = [latent_dim, 128, 64, 32, 32]
        for i in range(4):
            layers = [nn.ConvTranspose2d(widths[i], widths[i + 1], kernel_size=2, stride=2),
                      nn.BatchNorm2d(widths[i + 1]), nn.ReLU(inplace=True)]
            if i < 3:
                layers += _conv_bn_act(widths[i + 1], widths[i + 1], nn.ReLU(inplace=True))
            else:
                layers += [nn.Conv2d(widths[i + 1], out_channels, kernel_size=3, padding=1), nn.Tanh()]
            setattr(self, f"dec{i + 1}", nn.Sequential(*layers))

    def forward(self, x):
        for i in range(1, 5):
            x = getattr(self, f"dec{i}")(x)
        return x


def _init_like_reference(module: nn.Module) -> None:
    """Xavier-normal conv/convT weights, zero biases, identity BN (reference models/autoencoder.py:170-179)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.xavier_normal_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class _HipScorer:
    """Packed-weight cache + workspace for one model instance."""

    def __init__(self):
        self.key = None
        self.packed = None
        self.mode = hip.PREC_FP32
        self.ws = None

    @staticmethod
    def state_key(module: nn.Module):
        """(storage, version counter, device) of every state tensor.  In-place edits through `.data` / custom kernels do
        not bump the counter (call `invalidate_packed()` after those); inference-mode tensors have no counter at all."""
        def version(t):
            try:
                return t._version
            except RuntimeError:          # tensors created under torch.inference_mode() do not track versions
                return -1
        return tuple((t.data_ptr(), version(t), str(t.device)) for t in module.state_dict(keep_vars=True).values())

    @staticmethod
    def float_params(module: nn.Module):
        """State-dict tensors in order, num_batches_tracked dropped, as contiguous fp32 numpy."""
        out = []
        for k, v in module.state_dict().items():
            if k.endswith("num_batches_tracked"):
                continue
            out.append(np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy()))
        return out

    @staticmethod
    def widen_to_rgb(params: list, in_channels: int, last_transposed: bool) -> list:
        """The kernels read and reconstruct 3 planes (every reference call site passes in_channels=3, evaluate.py:35,
        evaluate_video.py:99).  A model built with 1 or 2 input channels is scored as the 3-channel model whose extra
        input taps and extra output channels are zero: first conv weight (32,C,3,3) -> (32,3,3,3), last layer (Conv2d
        weight (C,32,3,3) or ConvTranspose2d weight (32,C,2,2), bias (C)) -> 3 outputs.  The extra planes reconstruct
        tanh(0) = 0 against a zero input plane, so they add nothing to the squared error (the callers rescale the mean).
        Models with MORE than 3 channels are packed as they are (`kernel_channels`): the library runs their first and last
        layer on its generic kernels (csrc/wide_io.hip)."""
        if in_channels >= 3:
            return params
        params = list(params)
        w0 = params[0]
        params[0] = np.ascontiguousarray(np.concatenate([w0, np.zeros((w0.shape[0], 3 - in_channels, 3, 3), np.float32)], axis=1))
        wl, bl = params[-2], params[-1]
        if last_transposed:
            wl = np.concatenate([wl, np.zeros((wl.shape[0], 3 - in_channels) + wl.shape[2:], np.float32)], axis=1)
        else:
            wl = np.concatenate([wl, np.zeros((3 - in_channels,) + wl.shape[1:], np.float32)], axis=0)
        params[-2] = np.ascontiguousarray(wl)
        params[-1] = np.ascontiguousarray(np.concatenate([bl, np.zeros(3 - in_channels, np.float32)]))
        return params

    @staticmethod
    def kernel_channels(in_channels: int) -> int:
        """Planes the library reads and reconstructs for a model of `in_channels`: 3 for 1-3 (zero-widened), else as built."""
        return max(3, int(in_channels))

    @staticmethod
    def check_input(x: torch.Tensor, ndim: int, in_channels: int, what: str = "input") -> None:
        """Shape / dtype / device contract of a scoring call, checked BEFORE anything is packed or uploaded.  Float input
        is [..., C, H, W]; uint8 input [..., H, W, C] is offered for 3-channel models only (see `widen_input`)."""
        lead = "B," if ndim == 4 else "B,T,"
        if what != "input":
            lead = "F,"
        u8 = x.dtype == torch.uint8
        forms = f"float {what} [{lead}{in_channels},H,W]" + (f" or uint8 {what} [{lead}H,W,3]" if in_channels == 3 else "")
        if u8 and in_channels != 3:
            raise hip.VadError(f"uint8 {what} needs in_channels == 3 (this model has {in_channels}): expected {forms}")
        if x.dim() != ndim or (x.shape[-1] if u8 else x.shape[-3]) != in_channels:
            raise hip.VadError(f"expected {forms}, got {x.dtype} {tuple(x.shape)}")
        if not x.is_cuda:
            raise hip.VadError("inference runs only on the MI355X HIP path: move the model and input to 'cuda' "
                               "(there is no CPU fallback)")

    @staticmethod
    def widen_input(x: torch.Tensor, in_channels: int, u8: bool) -> torch.Tensor:
        """Zero planes appended to a 1- or 2-channel input (channel axis: last for uint8 NHWC, -3 for float NCHW)."""
        if in_channels >= 3:
            return x
        shape = list(x.shape)
        axis = x.dim() - 1 if u8 else x.dim() - 3
        shape[axis] = 3 - in_channels
        # uint8 frames are normalised as (u/255 - 0.5)/0.5 inside the kernels: the byte that maps closest to 0 does not
        # map to 0 exactly, so raw-byte input is offered for 3-channel models only
        if u8:
            raise hip.VadError("uint8 input needs in_channels == 3")
        return torch.cat([x, x.new_zeros(shape)], dim=axis)

    @staticmethod
    def warn_eval_with_grad(module: nn.Module) -> None:
        """An `eval()` model called with autograd enabled runs the stock torch.nn composition (a differentiable graph is
        what such a call asks for) - the one way to leave the HIP path without an error.  Every scoring call site of the
        reference is under `torch.no_grad()` (evaluate.py:56, evaluate_video.py:138,346, main.py:273,348); say so once per
        model instead of being silently slow."""
        if getattr(module, "_warned_eval_grad", False):
            return
        module._warned_eval_grad = True
        import warnings
        warnings.warn(f"{type(module).__name__} is in eval() mode but autograd is enabled: this call runs the torch.nn "
                      "composition, not the MI355X HIP path.  Wrap scoring in `torch.no_grad()` (as evaluate.py:56 does) "
                      "to use the HIP kernels.", RuntimeWarning, stacklevel=4)

    def workspace(self, nbytes: int, device) -> torch.Tensor:
        if self.ws is None or self.ws.numel() < nbytes or self.ws.device != device:
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.ws


class ConvAutoencoder(nn.Module):
    """Reference `ConvAutoencoder(in_channels=3, latent_dim=256)` (models/autoencoder.py:149-221)."""

    #: frames per launch group (workspace = 2 x chunk x 8.4 MB at 256x256 = 8 GiB of the 288): the whole 512-frame batch of
    #: configs[1] in one group - every launch has a ramp-up and a tail, and 16 launches per step instead of 64 measured +1.1 %
    #: (64 / 128 / 256 / 512 frames per group: 16.13 / 16.39 / 16.54 / 16.57 k frames/s)
    chunk = 512
    #: "fp32" = exact fp32 MFMA, direct convolutions (default, the parity path); "split" = 3 x fp16 MFMA with fp32 accumulate
    #: (opt-in, 22-bit products; include/vad_hip.h VAD_PREC_SPLIT); "winograd" = the 3x3 convolutions behind the first layer as
    #: Winograd F(2x2,3x3) on the exact-fp32 MFMA (opt-in: all-fp32 arithmetic, 16 instead of 36 products per 2x2 outputs, not
    #: bit-identical to "fp32"; VAD_PREC_WINO).  Per model: the mode travels with every call as an argument
    precision = "fp32"

    def __init__(self, in_channels: int = 3, latent_dim: int = 256):
        super().__init__()
        self.in_channels = in_channels
        self.latent_dim = latent_dim
        self.encoder = Encoder(in_channels, latent_dim)
        self.decoder = Decoder(in_channels, latent_dim)
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
            n = l.vad_img_packed_floats(kc, self.latent_dim) if 1 <= self.in_channels <= hip.MAX_IN_CHANNELS else 0
            if n == 0:
                raise hip.VadError(
                    f"ConvAutoencoder(in_channels={self.in_channels}, latent_dim={self.latent_dim}) is not "
                    f"supported by the HIP path (needs 1 <= in_channels <= {hip.MAX_IN_CHANNELS} and 1 <= latent_dim <= {hip.MAX_WIDTH})")
            params = _HipScorer.widen_to_rgb(_HipScorer.float_params(self), self.in_channels, last_transposed=False)
            blob = np.empty(n, dtype=np.float32)
            hip.check(l.vad_img_pack(hip.pointer_array(params), len(params), kc,
                                     self.latent_dim, mode, blob.ctypes.data), "vad_img_pack")
            self._hip.packed = torch.from_numpy(blob).to(device)
            self._hip.key = key
            self._hip.mode = mode        # the mode this blob was packed for: passed to every launch that reads it
        return self._hip.packed

    def invalidate_packed(self) -> None:
        """Drop the packed-weight cache.  Needed only after edits the cache key cannot see: writes through `p.data`
        (`p.data.copy_()`, EMA swaps), custom kernels writing into parameter storage.  `load_state_dict`, in-place ops on
        the parameters and optimiser steps are detected by themselves; `train()` / `eval()` drop the cache too."""
        self._hip.key = None

    def train(self, mode: bool = True):
        if mode != self.training:         # entering / leaving training: re-pack on the next scoring call
            self._hip.key = None
        return super().train(mode)

    def capture(self, x: torch.Tensor, scores=True, errmap=False, recon=False, latent=False) -> "hip.CapturedCall":
        """Capture ONE scoring call on a batch shaped like `x` into a hipGraph (include/vad_hip.h vad_graph_*) and return
        the replayable call: `g = model.capture(x); out = g.replay(new_x)` copies `new_x` into the captured input buffer and
        relaunches the whole layer sequence as one graph launch; `out` is the dict of captured output tensors (overwritten by
        every replay), bit-identical to the eager call.  Meant for the reference's call sizes (batch 16, evaluate.py:240),
        where the 16 launches of a call are short.  The graph reads the weights as packed NOW: capture again after
        changing parameters."""
        if not self._use_hip():
            raise hip.VadError("capture is an inference entry point: call under eval() and torch.no_grad()")
        want = dict(scores=scores, errmap=errmap, recon=recon, latent=latent)
        _HipScorer.check_input(x, 4, self.in_channels)
        # the captured kernels read and write 3 planes: a 1- / 2-channel batch is widened ONCE, outside the capture, into the
        # persistent buffer the graph reads (`replay(x)` copies into its first planes); the outputs are captured at kernel
        # shape and narrowed / rescaled after every replay.  (Widening inside the capture would record a `torch.cat` into
        # a temporary that is freed when the capture ends.)
        xs3 = _HipScorer.widen_input(x.contiguous() if x.dtype == torch.uint8 else x.contiguous().float(), self.in_channels,
                                     x.dtype == torch.uint8).clone()
        eager = self._run_hip(xs3, prewidened=True, **want)   # packs weights, sizes the workspace, creates helper state
        out = {k: torch.empty_like(v) for k, v in eager.items()}
        view = xs3 if self.in_channels >= 3 else xs3[:, :self.in_channels]
        return hip.CapturedCall(lambda: self._run_hip(xs3, out=out, prewidened=True, **want), view, out,
                                keep=(self._hip.packed, self._hip.ws, xs3),
                                post=None if self.in_channels >= 3 else self._narrow_outputs)

    def _run_hip(self, x: torch.Tensor, scores=False, errmap=False, recon=False, latent=False, out=None, prewidened=False):
        """`prewidened` (captured calls): `x` already has the kernels' planes and the outputs stay at kernel shape."""
        u8 = x.dtype == torch.uint8       # raw decoded frames [B,H,W,3]: normalised inside the kernels (row f-3)
        kc = _HipScorer.kernel_channels(self.in_channels)
        cin = kc if prewidened else self.in_channels
        _HipScorer.check_input(x, 4, cin)
        if u8:
            b, h, w, _ = x.shape
            x = x.contiguous()
        else:
            b, _, h, w = x.shape
            x = x.contiguous().float()
        l = hip.lib()
        dev = x.device
        packed = self._packed(dev)
        x = _HipScorer.widen_input(x, cin, u8)
        chunk = max(1, min(int(self.chunk), b))
        nbytes = l.vad_img_workspace_bytes_c(chunk, h, w, self.latent_dim, kc)
        if nbytes == 0:
            raise hip.VadError(f"unsupported frame size {h}x{w}: H and W must be multiples of 16")
        ws = self._hip.workspace(nbytes, dev)
        if out is None:                                   # (a captured call hands in its own output tensors)
            out = {}
            if scores:
                out["scores"] = torch.empty(b, dtype=torch.float32, device=dev)
            if errmap:
                out["errmap"] = torch.empty(b, 1, h, w, dtype=torch.float32, device=dev)
            if recon:
                out["recon"] = torch.empty(b, kc, h, w, dtype=torch.float32, device=dev)
            if latent:
                out["latent"] = torch.empty(b, self.latent_dim, h // 16, w // 16, dtype=torch.float32, device=dev)
        if b == 0:                                        # an empty batch gives empty outputs, as the reference's modules do
            return out
        with torch.cuda.device(dev):
            hip.check(l.vad_img_score_c(x.data_ptr(), hip.X_U8_NHWC if u8 else hip.X_F32_NCHW, self._hip.mode, kc, b, h, w,
                                        self.latent_dim, packed.data_ptr(), ws.data_ptr(),
                                        ws.numel(), chunk, hip.ptr(out.get("scores")), hip.ptr(out.get("errmap")),
                                        hip.ptr(out.get("recon")), hip.ptr(out.get("latent")), hip.current_stream()),
                      "vad_img_score")
        hip.calls["img_score"] += 1
        return out if prewidened else self._narrow_outputs(out)

    def _narrow_outputs(self, out: dict) -> dict:
        """Undo the 3-plane view of a 1- / 2-channel model (`_HipScorer.widen_to_rgb`): the kernels averaged over 3
        planes of which 3 - in_channels are exactly zero."""
        cin = self.in_channels
        if cin >= 3:
            return out
        out = dict(out)
        for k in ("scores", "errmap"):
            if k in out:
                out[k] = out[k] * (3.0 / cin)
        if "recon" in out:
            out["recon"] = out["recon"][:, :cin].contiguous()
        return out

    # ------------------------------------------------------------------ reference API
    def forward(self, x):
        """[B,C,H,W] -> reconstruction [B,C,H,W] (reference models/autoencoder.py:181-193)."""
        if self._use_hip():
            return self._run_hip(x, recon=True)["recon"]
        return self.decoder(self.encoder(x))

    def get_latent(self, x):
        """Latent code [B,latent,H/16,W/16] (reference models/autoencoder.py:195-197)."""
        if self._use_hip():
            return self._run_hip(x, latent=True)["latent"]
        return self.encoder(x)

    def get_reconstruction_error(self, x, per_pixel: bool = False):
        """Anomaly score: channel-mean squared error map [B,1,H,W] or its spatial mean [B]
        (reference models/autoencoder.py:199-221)."""
        if self._use_hip():
            if per_pixel:
                return self._run_hip(x, errmap=True)["errmap"]
            return self._run_hip(x, scores=True)["scores"]
        recon = self.decoder(self.encoder(x))
        error = ((x - recon) ** 2).mean(dim=1, keepdim=True)
        return error if per_pixel else error.mean(dim=[1, 2, 3])

    def score_all(self, x):
        """One pass returning recon, error map and scores together (the reference's callers run
        2-3 forwards for these: evaluate.py:146-147, main.py:274-276)."""
        if not self._use_hip():
            raise hip.VadError("score_all is an inference entry point: call under eval() and torch.no_grad()")
        return self._run_hip(x, scores=True, errmap=True, recon=True)


Autoencoder = ConvAutoencoder  # name used by BASELINE.json's north_star
