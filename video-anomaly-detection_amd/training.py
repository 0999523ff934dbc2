"""Native training step for `VideoAutoencoder`.

Replaces the body of the reference's `train_one_epoch` loop (train_video.py:50-60):

    reconstructions = model(sequences); loss = criterion(reconstructions, sequences)      # nn.MSELoss
    optimizer.zero_grad(); loss.backward(); optimizer.step()                              # Adam(lr, weight_decay=1e-5)

with `loss = trainer.step(sequences)`: forward in train() semantics (batch-statistics BatchNorm, running stats and
num_batches_tracked updated), MSE loss, the whole backward and the Adam update run as hand-written HIP kernels through
`vad_vid_train_fwd_bwd` + `vad_adam_step` (csrc/train_step.hip, csrc/train_ops.hip); there is no autograd graph and no
CPU fallback.  The model's `nn.Parameter`s and BatchNorm buffers are re-pointed to views of flat device buffers, so
`state_dict()`, `load_state_dict()`, checkpoints and the eval-mode scoring path keep working on the same storage, the
optimiser is one launch, and data-parallel training needs exactly one gradient all-reduce per step (RCCL through
torch.distributed; BatchNorm statistics stay per rank, the reference has no SyncBN).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import hip
from .video_autoencoder import VideoAutoencoder


def broadcast_(flat: torch.Tensor, src: int = 0, group=None, force_collective: bool = False) -> None:
    """Overwrite `flat` on every rank with rank `src`'s copy (same gloo host-copy route as `allreduce_sum_`).
    `force_collective`: issue the collective in a one-rank group too (exercises the RCCL call on a one-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        return
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)


def allreduce_sum_(flat: torch.Tensor, group=None, force_collective: bool = False) -> int:
    """Sum `flat` over the ranks of `group`, in place; returns the world size (1 when no process group is initialised).
    Backend "nccl" is RCCL on ROCm: one collective over the flat gradient buffer on the device.  With gloo (the CPU tests,
    single-GPU rehearsals with several ranks sharing one device) a GPU tensor goes through a host copy."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return 1
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return world


class _FlatTrainer:
    """Shared host logic of the native trainers: the module's parameters / BatchNorm buffers become views of flat device
    buffers (one optimiser launch, one gradient all-reduce), torch.optim.Adam semantics and state-dict format."""

    #: issue the gradient all-reduce in a one-rank process group too (tests: the RCCL call path on a one-GPU box)
    force_collective = False

    def __init__(self, model: nn.Module, nparams: int, nstats: int, lr, weight_decay, betas, eps, process_group):
        params = list(model.parameters())
        self.model, self.group = model, process_group
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.device = params[0].device
        if nparams == 0 or nparams != sum(p.numel() for p in params):
            raise hip.VadError(f"native training does not support this configuration ({sum(p.numel() for p in params)} "
                               f"parameters vs layout {nparams})")
        self.flat = torch.empty(nparams, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(nparams, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1).float())
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            off += k
        self.bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
        if nstats != sum(2 * m.num_features for m in self.bns):
            raise hip.VadError("unexpected BatchNorm layout")
        self.running = torch.empty(nstats, dtype=torch.float32, device=self.device)
        off = 0
        for m in self.bns:
            c = m.num_features
            self.running[off:off + c].copy_(m.running_mean)
            self.running[off + c:off + 2 * c].copy_(m.running_var)
            m.running_mean = self.running[off:off + c]
            m.running_var = self.running[off + c:off + 2 * c]
            off += 2 * c
        self.steps = 0
        self._ws: Optional[torch.Tensor] = None
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        # Data-parallel replicas must start equal (DistributedDataParallel broadcasts rank 0's module at construction;
        # the reference's train_video.py sets no seed, so independently constructed ranks would otherwise diverge silently).
        broadcast_(self.flat, 0, process_group)
        broadcast_(self.running, 0, process_group)

    def _check_aliasing(self) -> None:
        """The parameters must still be views of the flat buffers: `model.to()`, `.half()`, `.float()` or assigning
        `p.data` re-allocates them, after which the kernels would update storage the module no longer reads."""
        off = 0
        for p in self.model.parameters():
            if p.data_ptr() != self.flat.data_ptr() + 4 * off or p.dtype != torch.float32:
                raise hip.VadError("a parameter no longer aliases the trainer's flat buffer (model.to()/.half()/.float() or a "
                                   "p.data assignment after the trainer was built): build a new trainer from the module")
            off += p.numel()
        if self.bns and self.bns[0].running_mean.data_ptr() != self.running.data_ptr():
            raise hip.VadError("a BatchNorm buffer no longer aliases the trainer's flat buffer: build a new trainer from the module")

    @staticmethod
    def _check_model(model, cls, name):
        if not isinstance(model, cls):
            raise hip.VadError(f"{name} drives a {cls.__name__}")
        params = list(model.parameters())
        if not params or not all(p.is_cuda for p in params):
            raise hip.VadError(f"{name} needs the model on a GPU: model.cuda() first (there is no CPU fallback)")
        if model.in_channels != 3:
            raise hip.VadError(f"native training supports in_channels == 3 (got {model.in_channels})")

    def _ensure_ws(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _after_forward_backward(self, counter: str) -> None:
        for m in self.bns:
            m.num_batches_tracked += 1
        hip.calls[counter] = hip.calls.get(counter, 0) + 1


    def optimizer_step(self, grad_scale: float = 1.0) -> None:
        """torch.optim.Adam semantics (L2 weight decay added to the gradient) over the flat buffers."""
        self.steps += 1
        with torch.cuda.device(self.device):
            hip.check(hip.lib().vad_adam_step(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), self.flat.numel(), self.lr, self.betas[0], self.betas[1],
                                              self.eps, self.weight_decay, self.steps, float(grad_scale), hip.current_stream()),
                      "vad_adam_step")
        self.model._hip.key = None       # the eval-mode packed-weight cache is stale now

    # ------------------------------------------------------------------------------------------- checkpoint / resume
    def state_dict(self) -> dict:
        """Optimiser state in `torch.optim.Adam.state_dict()` format (what train_video.py:244,279 stores as
        'optimizer_state_dict'): loadable by a stock Adam over the same module, and vice versa."""
        params = list(self.model.parameters())
        state, off = {}, 0
        for i, p in enumerate(params):
            k = p.numel()
            if self.steps > 0:
                state[i] = {"step": torch.tensor(float(self.steps)),
                            "exp_avg": self.exp_avg[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).clone()}
            off += k
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        """Resume from a `torch.optim.Adam.state_dict()` (or this class's own): moments, step count and hyper-parameters."""
        groups = sd.get("param_groups", [])
        params = list(self.model.parameters())
        if len(groups) != 1 or list(groups[0].get("params", [])) != list(range(len(params))):
            raise hip.VadError("optimizer state does not describe one parameter group over this module's parameters")
        g = groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("decoupled_weight_decay"):
            raise hip.VadError("only plain Adam (no amsgrad / maximize / decoupled weight decay) is implemented")
        self.lr, self.betas = float(g["lr"]), tuple(float(b) for b in g["betas"])
        self.eps, self.weight_decay = float(g["eps"]), float(g["weight_decay"])
        st = sd.get("state", {})
        steps = {int(float(v["step"])) for v in st.values()}
        if len(steps) > 1 or (st and len(st) != len(params)):
            raise hip.VadError("per-parameter step counts differ: not a state this single-step-count optimiser can resume")
        self.steps = steps.pop() if steps else 0
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        off = 0
        for i, p in enumerate(params):
            k = p.numel()
            if st:
                e = st[i] if i in st else st[str(i)]
                if tuple(e["exp_avg"].shape) != tuple(p.shape):
                    raise hip.VadError(f"optimizer state of parameter {i} has shape {tuple(e['exp_avg'].shape)}, expected {tuple(p.shape)}")
                self.exp_avg[off:off + k].copy_(e["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(e["exp_avg_sq"].reshape(-1))
            off += k

    def step(self, batch: torch.Tensor) -> torch.Tensor:
        """One optimisation step on this rank's batch; with a process group the gradients are summed over ranks with ONE
        all-reduce of the flat buffer and averaged inside the optimiser kernel (DistributedDataParallel semantics)."""
        self._check_aliasing()
        loss, _ = self.forward_backward(batch)
        world = allreduce_sum_(self.grad, self.group, self.force_collective)
        self.optimizer_step(1.0 / world)
        return loss


class VideoTrainer(_FlatTrainer):
    """Adam-on-MSE training steps for a `VideoAutoencoder` living on a GPU (exact fp32, or split-fp16 convolutions)."""

    def __init__(self, model: VideoAutoencoder, lr: float = 1e-4, weight_decay: float = 1e-5, betas=(0.9, 0.999),
                 eps: float = 1e-8, process_group=None, precision: str = "fp32"):
        self._check_model(model, VideoAutoencoder, "VideoTrainer")
        #: "fp32": exact fp32 everywhere (default, the parity path).  "split": the 3x3 / transposed convolutions (forward
        #: and data gradients) use split-fp16 operands - 22-bit products, fp32 accumulate - everything else stays fp32.
        #: "bf16" (BASELINE.json configs[4]) = "bf16_tensors": every activation and activation-gradient tensor between the
        #: kernels is bf16 in HBM and every convolution / weight-gradient GEMM behind the first layer runs on bf16 MFMA
        #: operands with fp32 accumulation; arithmetic inside the kernels, BatchNorm statistics, cell states, master weights,
        #: parameter gradients, loss and Adam stay fp32.  "bf16_operands" (round 2's form, kept for A/B): fp32 tensors in
        #: HBM, converted to bf16 while they are staged.  Both are gated by loss-curve agreement, not parity
        self.precision = hip.training_precision(precision, "VideoTrainer", tensors=True)
        l = hip.lib()
        self.cfg = (model.latent_dim, model.lstm_hidden_dim, model.lstm_num_layers)
        super().__init__(model, l.vad_vid_train_nparams(*self.cfg), l.vad_vid_train_nstats(*self.cfg), lr, weight_decay, betas, eps,
                         process_group)

    # ------------------------------------------------------------------------------------------------------------
    def _workspace(self, b, t, h, w) -> torch.Tensor:
        nbytes = hip.lib().vad_vid_train_workspace_bytes(b, t, h, w, *self.cfg)
        if nbytes == 0:
            raise hip.VadError(f"unsupported training shape B={b} T={t} {h}x{w}: H and W must be multiples of 16")
        return self._ensure_ws(nbytes)

    def forward_backward(self, clips: torch.Tensor, recon: bool = False):
        """Loss and gradients of one batch (train-mode forward, MSE, full backward) without the optimiser update.
        Returns (loss 0-dim device tensor, reconstruction or None); gradients are in `p.grad` of every parameter."""
        if clips.dim() != 5 or clips.shape[2] != 3 or not clips.is_cuda:
            raise hip.VadError(f"expected GPU clips [B,T,3,H,W], got {tuple(clips.shape)} on {clips.device}")
        x = clips.contiguous().float()
        b, t, _, h, w = x.shape
        ws = self._workspace(b, t, h, w)
        out = torch.empty_like(x) if recon else None
        l = hip.lib()
        with torch.cuda.device(self.device):
            hip.check(l.vad_vid_train_fwd_bwd(x.data_ptr(), b, t, h, w, *self.cfg, self.flat.data_ptr(), self.grad.data_ptr(),
                                              self.running.data_ptr(), ws.data_ptr(), ws.numel(), hip.precision_mode(self.precision),
                                              self._loss.data_ptr(), hip.ptr(out), hip.current_stream()), "vad_vid_train_fwd_bwd")
        self._after_forward_backward("train_step")
        return self._loss[0].clone(), out


class ImageTrainer(_FlatTrainer):
    """Native training step for `ConvAutoencoder` (reference train.py:28-52): `loss` is 'mse' (train.py's default),
    'ssim' or 'combined' (train.py:149-158, `--ssim-weight` = alpha); Adam(lr 1e-3, weight_decay 1e-5) as in train.py:159."""

    _KINDS = {"mse": 0, "ssim": 1, "combined": 2}

    def __init__(self, model, lr: float = 1e-3, weight_decay: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, loss: str = "mse", ssim_weight: float = 0.5, window_size: int = 11, precision: str = "fp32"):
        from .autoencoder import ConvAutoencoder
        self._check_model(model, ConvAutoencoder, "ImageTrainer")
        #: "fp32" | "split" | "winograd" (fp32 everywhere, 3x3 forward / data-gradient convolutions as Winograd F(2x2,3x3)) |
        #: "bf16_operands" (fp32 tensors, bf16 MFMA operands in the convolutions and weight gradients).
        #: "bf16" / "bf16_tensors" name the bf16-TENSOR mode of `VideoTrainer` and are rejected here by name: the image
        #: step has no such form (round 3 accepted "bf16" here with the operand meaning - the same string, other arithmetic)
        self.precision = hip.training_precision(precision, "ImageTrainer", tensors=False)
        if loss not in self._KINDS:
            raise hip.VadError(f"loss must be one of {sorted(self._KINDS)}, got {loss!r}")
        self.loss, self.ssim_weight, self.window_size = loss, float(ssim_weight), int(window_size)
        l = hip.lib()
        self.latent = model.latent_dim
        super().__init__(model, l.vad_img_train_nparams(self.latent), l.vad_img_train_nstats(self.latent), lr, weight_decay, betas, eps,
                         process_group)

    def forward_backward(self, images: torch.Tensor, recon: bool = False):
        if images.dim() != 4 or images.shape[1] != 3 or not images.is_cuda:
            raise hip.VadError(f"expected GPU images [B,3,H,W], got {tuple(images.shape)} on {images.device}")
        x = images.contiguous().float()
        b, _, h, w = x.shape
        l = hip.lib()
        nbytes = l.vad_img_train_workspace_bytes(b, h, w, self.latent)
        if nbytes == 0:
            raise hip.VadError(f"unsupported training shape B={b} {h}x{w}: H and W must be multiples of 16")
        ws = self._ensure_ws(nbytes)
        out = torch.empty_like(x) if recon else None
        with torch.cuda.device(self.device):
            hip.check(l.vad_img_train_fwd_bwd(x.data_ptr(), b, h, w, self.latent, self.flat.data_ptr(), self.grad.data_ptr(),
                                              self.running.data_ptr(), ws.data_ptr(), ws.numel(), self._KINDS[self.loss],
                                              self.ssim_weight, self.window_size, hip.precision_mode(self.precision),
                                              self._loss.data_ptr(), hip.ptr(out), hip.current_stream()), "vad_img_train_fwd_bwd")
        self._after_forward_backward("train_step_img")
        return self._loss[0].clone(), out

