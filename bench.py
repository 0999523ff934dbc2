"""bench.py — frames/s through the anomaly-scoring hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of get_reconstruction_error over one batch already resident in HBM:
  image (default; BASELINE.json configs[1]): 512 synthetic 256x256x3 frames per GPU;
  video (--workload video; configs[2]): 64 clips of 10 frames per GPU.
With N > 1 the frame stream is block-partitioned over ranks (weak scaling, no data-path collective) and each
step ends with the one all_gather of the score vector (RCCL).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
PEAK_HBM_GBS = 8000.0

# Algorithmic work per 256x256 frame (scaled by H*W/65536 otherwise)
IMG_FLOP_PER_FRAME = 8_111_783_936
IMG_BYTES_PER_FRAME = 72_351_748
VID_FLOP_PER_FRAME = 3_011_510_272      # at T = 10
VID_BYTES_PER_FRAME = 17_825_796


def conv3x3_flops(h, w, cin, cout):
    return 2.0 * h * w * cin * cout * 9


def image_mfma_layer_flops(h, w, latent):
    """slot -> algorithmic FLOP per frame of the conv3x3 MFMA launches (vad_api.hip slot numbering)."""
    ch = [3, 32, 64, 128, latent]
    f = {}
    hh, ww = h, w
    f[1] = conv3x3_flops(hh, ww, 32, 32) + conv3x3_flops(hh, ww, 3, 32)   # fused enc1.0 + enc1.3 launch
    for blk in range(1, 4):
        hh, ww = hh // 2, ww // 2
        f[2 * blk] = conv3x3_flops(hh, ww, ch[blk], ch[blk + 1])
        f[2 * blk + 1] = conv3x3_flops(hh, ww, ch[blk + 1], ch[blk + 1])
    hh, ww = hh // 2, ww // 2
    dch = [latent, 128, 64, 32]
    for blk in range(3):
        hh, ww = hh * 2, ww * 2
        f[9 + 2 * blk] = conv3x3_flops(hh, ww, dch[blk + 1], dch[blk + 1])
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["image", "video", "dense"], default="image")
    ap.add_argument("--stream-frames", type=int, default=-1,
                    help="configs[3]: also time scoring.score_stream over this many frames (image workload, reported as `stream`); "
                         "-1 = 100000 in the default line (image, fp32, default batch / size), 0 = skip")
    ap.add_argument("--no-video", action="store_true", help="skip the secondary configs[2] measurement of the default line")
    ap.add_argument("--stride", type=int, default=1, help="dense workload: window stride")
    ap.add_argument("--precision", choices=["fp32", "split", "winograd"], default="fp32",
                    help="fp32 = exact fp32 MFMA, direct convolutions (headline); split = opt-in 3 x fp16 MFMA with fp32 accumulate; "
                         "winograd = opt-in Winograd F(2x2,3x3) on the exact-fp32 MFMA")
    ap.add_argument("--ingest", choices=["f32", "u8"], default="f32",
                    help="row f-3: u8 = raw uint8 NHWC frames, normalised inside the kernels (image workload)")
    ap.add_argument("--batch", type=int, default=0, help="frames (image) or clips (video) per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="frames/clips per launch group (0 = model default)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--clip-len", type=int, default=10)
    ap.add_argument("--no-layer-events", action="store_true", help="do not bracket layers with hipEvents")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the secondary split-precision measurement")
    ap.add_argument("--no-small", action="store_true", help="skip the secondary measurement of the reference's own call sizes")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement (row f-1)")
    ap.add_argument("--graph", action="store_true", help="capture the step's scoring call into a hipGraph once and replay it every step (1 GPU, image / video)")
    ap.add_argument("--no-wavefront", action="store_true", help="video: ConvLSTM layers strictly one after the other (A/B of the small-batch wavefront)")
    ap.add_argument("--always-wavefront", action="store_true", help="video: ConvLSTM layer wavefront at any batch size (A/B)")
    ap.add_argument("--conv-variant", type=int, default=-1, help="vad_debug_set_conv_variant bits (A/B; 9 = never the small-grid ConvLSTM kernel)")
    ap.add_argument("--tail-group", type=int, default=0, help="frames per dec4.0 -> tail sub-group (0 = auto)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use GPU 0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the scoring path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    vad = importlib.import_module("video-anomaly-detection_amd")
    hip = vad.hip
    lib = hip.lib()
    hw = args.size
    scale = (hw * hw) / 65536.0
    if args.tail_group:
        lib.vad_debug_set_tail_group(args.tail_group)
    if args.no_wavefront:
        lib.vad_debug_set_lstm_wavefront(0)
    if args.always_wavefront:
        lib.vad_debug_set_lstm_wavefront(2)
    if args.conv_variant >= 0:
        lib.vad_debug_set_conv_variant(args.conv_variant)

    def synth_load(module, seed):
        shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
        st = {k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, seed).items()}
        module.load_state_dict(st, strict=True)
        return st

    seed = 0xC0FFEE + (1 if args.workload == "image" else 2)
    if args.workload == "image":
        per_gpu = args.batch or 512
        model = vad.ConvAutoencoder(in_channels=3, latent_dim=256)
        state = synth_load(model, 7)
        model = model.to(dev).eval()
        if args.chunk:
            model.chunk = args.chunk
        x = vad.scoring.synth_frames_device(seed, rank * per_gpu, per_gpu, hw, hw, device=dev)
        if args.ingest == "u8":   # the same frames as uint8 NHWC (what a decoder hands over): 4x fewer input bytes
            u8 = vad.synth.frames_u8(seed, rank * per_gpu, per_gpu, 3, hw, hw)
            x = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).to(dev)
        frames_per_step = per_gpu
        flop_per_frame, bytes_per_frame = IMG_FLOP_PER_FRAME * scale, IMG_BYTES_PER_FRAME * scale

        def score_block(first, count):
            return model.get_reconstruction_error(x)
        width = 1
        workload = f"configs[1]: image autoencoder scoring, batch {per_gpu} synthetic {hw}x{hw}x3 frames per GPU"
        if args.ingest == "u8":
            workload += " (row f-3: uint8 NHWC ingest, normalised in-kernel)"
    elif args.workload == "dense":
        # Row f-2: dense sliding windows over ONE video per GPU (reference evaluate_video.py:322-352 with
        # sequence_length T, stride): per_gpu windows, every frame encoded once.
        per_gpu = args.batch or 512
        t = args.clip_len if args.clip_len != 10 else 16
        nfr = (per_gpu - 1) * args.stride + t
        model = vad.VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128, lstm_num_layers=2)
        state = synth_load(model, 8)
        model = model.to(dev).eval()
        if args.chunk:
            model.window_chunk = args.chunk
        x = vad.scoring.synth_frames_device(seed, rank * nfr, nfr, hw, hw, device=dev)
        frames_per_step = per_gpu * t
        flop_per_frame, bytes_per_frame = VID_FLOP_PER_FRAME * scale, VID_BYTES_PER_FRAME * scale

        def score_block(first, count):
            return model.score_windows(x, sequence_length=t, stride=args.stride)["frame"]
        width = t
        workload = (f"row f-2: dense sliding windows, {per_gpu} windows of {t} frames (stride {args.stride}) over one "
                    f"{nfr}-frame {hw}x{hw}x3 video per GPU; value counts window-frames (windows x T)")
    else:
        per_gpu = args.batch or 64
        t = args.clip_len
        model = vad.VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128, lstm_num_layers=2)
        state = synth_load(model, 8)
        model = model.to(dev).eval()
        if args.chunk:
            model.chunk = args.chunk
        x = vad.scoring.synth_frames_device(seed, rank * per_gpu * t, per_gpu * t, hw, hw, device=dev).view(per_gpu, t, 3, hw, hw)
        frames_per_step = per_gpu * t
        flop_per_frame, bytes_per_frame = VID_FLOP_PER_FRAME * scale, VID_BYTES_PER_FRAME * scale

        def score_block(first, count):
            return model.get_reconstruction_error(x, per_frame=True)
        width = t
        workload = f"configs[2]: ConvLSTM video autoencoder scoring, {per_gpu} clips x {t} frames of {hw}x{hw}x3 per GPU"

    model.precision = args.precision
    n_items = per_gpu * world
    if args.graph:
        if world != 1 or args.workload == "dense":
            raise SystemExit("--graph: 1 GPU, image or video workload")
        with torch.no_grad():
            captured = model.capture(x, scores=True) if args.workload == "image" else model.capture(x, seq=False, frame=True)
        key = "scores" if args.workload == "image" else "frame"

        def score_block(first, count):                     # noqa: F811 - one graph launch instead of the launch sequence
            return captured.replay()[key]
        workload += "; the call is captured into a hipGraph once and replayed (vad_graph_*)"

    def step():
        with torch.no_grad():
            return vad.scoring.sharded_scores(score_block, n_items, width=width, rank=rank, world=world, device=dev)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        scores = step()
    events = not args.no_layer_events
    fence()
    if events:
        lib.vad_prof_reset()
        lib.vad_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scores = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.vad_prof_enable(0)

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    total_frames = frames_per_step * world * args.steps
    fps = total_frames / elapsed

    layers, roofline = None, None
    if events:
        default_shape = hw == 256 and not args.batch and int(model.chunk) == (512 if args.workload == "image" else 64) and (args.workload != "video" or t == 10)
        layers, roofline = layers_and_roofline(hip, lib, args.workload, hw, per_gpu, args.steps, t if args.workload != "image" else 0,
                                               args.stride, default_shape, fps / world, flop_per_frame, bytes_per_frame)
        if args.precision == "winograd" and args.workload != "dense":
            # the timed path is the Winograd mode: its roofline counts EXECUTED matrix FLOPs (16/36 of the direct-convolution
            # FLOPs), or `frac` would exceed 1
            roofline = winograd_roofline(hip, lib, args.workload, hw, per_gpu, args.steps, t if args.workload != "image" else 0)["roofline"]

    out = {
        "metric": "frames/sec/GPU (256x256 autoencoder scoring) + AUROC parity vs reference",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32", "split": "f32 via 3 x f16 split MFMA (opt-in)",
                                       "winograd": "f32, Winograd F(2x2,3x3) 3x3 convolutions (opt-in)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu_per_step": frames_per_step,
                   "chunk": int(model.window_chunk if args.workload == "dense" else model.chunk),
                   "weights": "deterministic synthetic state dict (Xavier scale, randomised BN)",
                   "collective": "one all_gather of the score vector per step" if world > 1 else "none (1 GPU)"},
        "frames_per_sec_per_gpu": round(fps / world, 1),
        "roofline": roofline,
    }
    if layers is not None:
        out["layers"] = layers

    # Secondary measurements, same workload and step count: the opt-in arithmetic modes - split-fp16 (3 x fp16 MFMA, fp32
    # accumulate) and Winograd F(2x2,3x3) on the exact-fp32 MFMA (all-fp32, 16 instead of 36 products
    # per 2x2 outputs).  `value` above is always the exact-fp32 direct path.
    if args.precision == "fp32" and not args.no_split:
        exact_scores = scores.clone()
        for mode in ("split", "winograd"):
            if mode == "winograd" and args.workload == "dense":
                continue
            model.precision = mode
            for _ in range(max(1, args.warmup)):
                scores = step()
            fence()
            if mode == "winograd" and events:
                lib.vad_prof_reset()
                lib.vad_prof_enable(1)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                scores = step()
            fence()
            el2 = time.perf_counter() - t1
            lib.vad_prof_enable(0)
            if dist is not None:
                tt = torch.tensor([el2], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el2 = float(tt.item())
            diff = float(((scores - exact_scores).abs() / exact_scores.abs()).max())
            res = {"value": round(total_frames / el2, 1), "unit": "frames/s", "ms_per_step": round(el2 / args.steps * 1e3, 3),
                   "max_rel_score_diff_vs_exact_fp32": diff, "speedup_vs_direct": round(elapsed / el2, 3)}
            if mode == "split":
                res["arithmetic"] = "a*b = ah*bh + (ah*bl + al*bh)*2^-11, fp16 hi/lo operands, fp32 accumulate"
            else:
                res["arithmetic"] = ("Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32: every 3x3 convolution behind the first layer computes 2x2 outputs "
                                     "from 16 products per input channel instead of 36; all-fp32, another rounding order than the direct form")
                if events:
                    res.update(winograd_roofline(hip, lib, args.workload, hw, per_gpu, args.steps, t if args.workload != "image" else 0))
            out[mode + "_precision"] = res
        model.precision = "fp32"
        scores = exact_scores
    # configs[3]: the frame stream generated on the device in chunks of 512, block-partitioned over the ranks, ONE all_gather
    # at the end; generation is inside the timed region.  Part of the default line (image, fp32, default batch and size: the
    # driver's `bench.py --gpus N` run); 100,000 frames in total whatever the rank count (strong scaling, its own object).
    default_line = args.workload == "image" and args.precision == "fp32" and not args.batch and hw == 256 and args.ingest == "f32" and not args.graph
    stream_frames = args.stream_frames if args.stream_frames >= 0 else (100000 if default_line else 0)
    if stream_frames > 0 and args.workload == "image" and args.precision == "fp32":
        fence()
        t1 = time.perf_counter()
        sv = vad.scoring.score_stream(model, seed + 2, stream_frames, chunk=512, h=hw, w=hw, rank=rank, world=world, device=dev)
        fence()
        el3 = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([el3], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el3 = float(tt.item())
        out["stream"] = {"frames": stream_frames, "seconds": round(el3, 4), "value": round(stream_frames / el3, 1), "unit": "frames/s",
                         "scaling": "strong", "frames_per_rank": -(-stream_frames // world),
                         "config": {"workload": f"configs[3]: {stream_frames}-frame synthetic {hw}x{hw}x3 stream, block-partitioned over {world} rank(s), "
                                                "generated on the device in chunks of 512, one all_gather of the score vector at the end"},
                         "includes": "on-device frame generation, scoring, one all_gather", "checksum": float(sv.double().sum())}
        if rank == 0 and not args.no_cpu_baseline:
            # the GATHERED stream vector against the CPU oracle on one frame out of every rank's block (its first, a middle
            # one, or - last rank - the stream's very last frame): a wrong block, a wrong order or a lost tail fails here
            from oracle import torch_oracle
            torch.set_num_threads(host_cores())
            per = -(-stream_frames // world)
            picks = sorted({min(stream_frames - 1, r * per + (0 if r == 0 else per // 2)) for r in range(world)} | {stream_frames - 1})
            xs = torch.from_numpy(np.concatenate([vad.synth.frames(seed + 2, i, 1, 3, hw, hw) for i in picks]))
            with torch.no_grad():
                ref = torch_oracle.img_scores(state, xs)["scores"]
            rel = float(((sv[picks].cpu() - ref).abs() / ref.abs()).max())
            out["stream"]["parity"] = {"checked_frames": picks, "oracle": "oracle/torch_oracle.py on rank 0's host cores",
                                       "max_rel_score_err_vs_cpu": rel, "within_1e-4": bool(rel < 1e-4)}
            if not rel < 1e-4:
                raise SystemExit(f"stream parity check failed: scores differ from the CPU oracle by {rel:.3e} on frames {picks}")
    # configs[2] beside it (1 GPU: the driver's N = 1 line then carries every single-GPU configuration)
    if rank == 0 and world == 1 and default_line and not args.no_video:
        out["video"] = video_config2(vad, hip, lib, dev, hw, args.steps, args.warmup, not args.no_cpu_baseline)
    # The reference's own call sizes (main.py:274 one image, evaluate.py:240 16 images, evaluate_video.py:416 4 clips x 16 frames,
    # :344 one 16-frame window), eager launches, per-layer events off.  Never part of `value`.
    if rank == 0 and world == 1 and default_line and not args.no_small:
        out["reference_call_sizes"] = reference_call_sizes(vad, lib, model, dev, hw, seed)
        if "winograd_precision" in out:       # the same calls in the opt-in Winograd mode (never part of `value`)
            out["winograd_precision"]["reference_call_sizes"] = reference_call_sizes(vad, lib, model, dev, hw, seed, precision="winograd")
    # Secondary measurement (row f-1): the native training step of the ConvLSTM video autoencoder (train_video.py:44-65),
    # exact fp32, 32 clips x 10 frames at the bench resolution.  Never part of `value` / `roofline`.
    if rank == 0 and world == 1 and not args.no_train and args.workload == "image" and args.precision == "fp32":
        out["training_step"] = training_step(vad, dev, hw)
    if args.workload == "dense":
        out["unique_frames_per_sec"] = round(((per_gpu - 1) * args.stride + t) * world * args.steps / elapsed, 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != "dense":
        out["cpu_baseline"] = cpu_baseline(vad, state, args.workload, args.clip_len, scores, seed, hw)
    if world > 1:
        out["multi_gpu"] = multi_gpu_evidence(vad, dist, args, state, scores, seed, hw, per_gpu, width, rank, world, dev, local_rank)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def layers_and_roofline(hip, lib, kind, hw, per_gpu, steps, t, stride, default_shape, fps_per_gpu, flop_per_frame, bytes_per_frame):
    """Per-layer hipEvent times recorded by the library on the launch stream during the timed region (vad_prof_*), and the
    `roofline` object of the dominant kernel: achieved = sum of the algorithmic FLOP of its launches / sum of their durations."""
    import ctypes as C
    ms = (C.c_float * hip.PROF_SLOTS)()
    cnt = (C.c_int * hip.PROF_SLOTS)()
    hip.check(lib.vad_prof_read(ms, cnt), "vad_prof_read")
    model_id = 0 if kind == "image" else 1
    layers = {lib.vad_prof_slot_name(model_id, i).decode(): {"ms": round(ms[i], 4), "launches": cnt[i]}
              for i in range(hip.PROF_SLOTS) if cnt[i]}
    frames_step = per_gpu * (t if kind != "image" else 1)
    frames_rank = frames_step * steps
    if kind == "image":
        lf = image_mfma_layer_flops(hw, hw, 256)
        mf_ms = sum(ms[i] for i in lf)
        mf_launch = sum(cnt[i] for i in lf)
        mf_flop = sum(lf.values()) * frames_rank
        for i, fl in lf.items():
            nm = lib.vad_prof_slot_name(0, i).decode()
            layers[nm]["tflops"] = round(fl * frames_rank / (ms[i] * 1e-3) / 1e12, 2) if ms[i] > 0 else None
    else:
        # dominant kernel of the video path: the same conv3x3 MFMA kernel in its ConvLSTM form
        h16 = hw // 16
        step_flop = conv3x3_flops(h16, h16, 256, 512)          # per clip per (layer, t) launch
        enc = [conv3x3_flops(hw // 2, hw // 2, 32, 64), conv3x3_flops(hw // 4, hw // 4, 64, 128),
               conv3x3_flops(hw // 8, hw // 8, 128, 128)]
        mf_ms = ms[4] + ms[1] + ms[2] + ms[3]
        mf_launch = cnt[4] + cnt[1] + cnt[2] + cnt[3]
        # ConvLSTM FLOPs actually executed: at t = 0 the state is exactly zero and the h half of K is skipped
        # (the reference multiplies by zeros there), so a clip costs 2 layers x (T - 1/2) full steps.
        clips_rank = per_gpu * steps
        lstm_flop = step_flop * 2 * (t - 0.5) * clips_rank
        # frames that pass through the encoder: every (clip, t) frame, or each source frame once for dense windows
        enc_frames = ((per_gpu - 1) * stride + t) * steps if kind == "dense" else frames_rank
        mf_flop = lstm_flop + sum(enc) * enc_frames
        layers["convlstm"]["tflops"] = round(lstm_flop / (ms[4] * 1e-3) / 1e12, 2) if ms[4] > 0 else None
        # the same figure for the whole path (review of round 2: the frame constant counts the h half at t = 0 too)
        flop_per_frame = flop_per_frame - step_flop * 2 * 0.5 / t
    ach = mf_flop / (mf_ms * 1e-3) / 1e12 if mf_ms > 0 else 0.0
    # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    # (tools/pmc_traffic.sh: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); only valid for the default shape.
    # The file records the sha256 of the kernel sources it was measured on (hip.source_digest): when a kernel has changed
    # since, the figure is stale and `traffic` is null with the reason - never a number from other code.
    traffic, traffic_source = None, None
    pmcs = sorted((REPO / "profiles").glob(f"r*_pmc_traffic_{kind}.json"))     # newest round last
    if pmcs and default_shape:
        rec = json.loads(pmcs[-1].read_text())
        if rec.get("source_sha256") == hip.source_digest():
            traffic = round(rec["traffic_bytes_per_launch"])
            traffic_source = f"profiles/{pmcs[-1].name} (separate rocprofv3 --pmc passes of this command on these kernel sources, not this run)"
        else:
            traffic_source = (f"null: profiles/{pmcs[-1].name} was measured on other kernel sources (its source_sha256 "
                              f"{str(rec.get('source_sha256'))[:12]} != {hip.source_digest()[:12]}); re-run tools/pmc_traffic.sh")
    elif not default_shape:
        traffic_source = "null: the committed counter passes cover the default shape only"
    roofline = {"bound": "mfma", "kernel": "conv3x3_mfma_pkernel (fp32 32x32x2 MFMA; all launches)",
                "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_flop_per_launch": round(mf_flop / max(mf_launch, 1)),
                "avg_launch_ms": round(mf_ms / max(mf_launch, 1), 4), "launches": mf_launch,
                "whole_path_tflops": round(fps_per_gpu * flop_per_frame / 1e12, 2),
                "whole_path_hbm_gbs": round(fps_per_gpu * bytes_per_frame / 1e9, 1),
                "whole_path_hbm_frac": round(fps_per_gpu * bytes_per_frame / 1e9 / PEAK_HBM_GBS, 4)}
    return layers, roofline


def winograd_roofline(hip, lib, kind, hw, per_gpu, steps, t):
    """`roofline` of the Winograd launches of the timed region just recorded (vad_prof_*): EXECUTED matrix FLOPs (16/36 of a
    layer's direct-convolution FLOPs) against the fp32 MFMA peak, so `frac` <= 1; the direct-equivalent rate beside it."""
    import ctypes as C
    ms = (C.c_float * hip.PROF_SLOTS)()
    cnt = (C.c_int * hip.PROF_SLOTS)()
    hip.check(lib.vad_prof_read(ms, cnt), "vad_prof_read")
    if kind == "image":
        lf = image_mfma_layer_flops(hw, hw, 256)
        lf[1] = conv3x3_flops(hw, hw, 32, 32)                   # enc1.3 alone: the first layer is its own (direct) launch in this mode
        frames = per_gpu * steps
        names = {i: lib.vad_prof_slot_name(0, i).decode() for i in lf}
    else:
        # slot 4: the ConvLSTM steps (gate convolution in Winograd form + the pointwise cell launch); per frame: 2 layers x one
        # 256 -> 512 convolution on the 16x16 map, the h half of K skipped at t = 0
        h16 = hw // 16
        lf = {1: conv3x3_flops(hw // 2, hw // 2, 32, 64), 2: conv3x3_flops(hw // 4, hw // 4, 64, 128), 3: conv3x3_flops(hw // 8, hw // 8, 128, 128),
              4: conv3x3_flops(h16, h16, 256, 512) * 2 * (t - 0.5) / t}
        frames = per_gpu * t * steps
        names = {i: lib.vad_prof_slot_name(1, i).decode() for i in lf}
    w_ms = sum(ms[i] for i in lf)
    direct_flop = sum(lf.values()) * frames
    layers = {names[i]: {"ms": round(ms[i] / steps, 4),
                         "direct_equivalent_tflops": round(lf[i] * frames / (ms[i] * 1e-3) / 1e12, 1) if ms[i] > 0 else None} for i in lf}
    ach = direct_flop * 16.0 / 36.0 / (w_ms * 1e-3) / 1e12 if w_ms > 0 else 0.0
    model_id = 0 if kind == "image" else 1
    for i in range(hip.PROF_SLOTS):          # the launches of the step that are not Winograd convolutions, for the whole picture
        nm = lib.vad_prof_slot_name(model_id, i).decode()
        if cnt[i] and nm not in layers:
            layers[nm] = {"ms": round(ms[i] / steps, 4)}
    for i in lf:
        layers[names[i]]["ms"] = round(ms[i] / steps, 4)
    return {"roofline": {"bound": "mfma", "kernel": "conv3x3_wino_pkernel (fp32 32x32x2 MFMA; all launches)", "achieved": round(ach, 2),
                         "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": None,
                         "flops": "executed matrix FLOPs = 16/36 of the direct-convolution FLOPs of these layers",
                         "direct_equivalent_tflops": round(direct_flop / (w_ms * 1e-3) / 1e12, 2) if w_ms > 0 else None,
                         "launches": sum(cnt[i] for i in lf), "ms": round(w_ms / steps, 3)},
            "layers": layers}


# Algorithmic work of the kernel groups of one training step (vad_prof_slot_name(2, .)), per 256x256 frame of a T = 10 clip of
# the default VideoAutoencoder (SURVEY.md Appendix B layer table; every layer's forward FLOPs recur once in its data gradient
# and once in its weight gradient).  GFLOP per frame for the MFMA groups; MB per frame of fp32 tensors for the HBM-bound
# passes (bf16 tensors: half): BatchNorm forward reads the conv outputs (15.2 encoder + 3.7 decoder) and writes the activated
# (pooled) maps (7.5); backward pass A reads conv outputs + upstream gradients, pass B the same plus the write of dy.
TRAIN_GROUP_GFLOP = {"first layer fwd": 0.11325, "conv3x3 fwd": 1.50995, "ConvLSTM conv fwd": 1.20796, "convT / proj fwd": 0.16777,
                     "conv3x3 dgrad": 1.50995, "ConvLSTM conv dgrad": 1.20796, "convT / proj dgrad (1x1)": 0.16777,
                     "weight gradients": 1.50995 + 1.20796 + 0.16777 + 0.01258, "first layer wgrad": 0.11325}
TRAIN_GROUP_MB_FP32 = {"BatchNorm fwd": 18.9 + 7.5, "BatchNorm bwd": (18.9 + 7.5) + (18.9 + 7.5 + 18.9)}
TRAIN_PEAK_TFLOPS = {"fp32": 157.3, "split": 2500.0 / 3, "bf16": 2500.0, "winograd": 157.3}


def training_groups(vad, frames, hw, precision):
    """Per-group hipEvent times of the step(s) recorded since the last vad_prof_reset -> {group: ms per step, achieved rate}."""
    import ctypes as C
    hip, lib = vad.hip, vad.hip.lib()
    ms = (C.c_float * hip.PROF_SLOTS)()
    cnt = (C.c_int * hip.PROF_SLOTS)()
    hip.check(lib.vad_prof_read(ms, cnt), "vad_prof_read")
    scale = (hw / 256.0) ** 2
    groups = {}
    for i in range(hip.PROF_SLOTS):
        if not cnt[i]:
            continue
        name = lib.vad_prof_slot_name(2, i).decode()
        g = {"ms": round(ms[i], 4), "launches": cnt[i]}
        if name in TRAIN_GROUP_GFLOP and ms[i] > 0:
            g["tflops"] = round(TRAIN_GROUP_GFLOP[name] * scale * frames / ms[i], 2)        # GFLOP / ms = TFLOP/s
        if name in TRAIN_GROUP_MB_FP32 and ms[i] > 0:
            mb = TRAIN_GROUP_MB_FP32[name] * scale * (0.5 if precision == "bf16" else 1.0)
            g["algorithmic_GB"] = round(mb * frames / 1e3, 3)
            g["TBps"] = round(mb * frames / ms[i] / 1e3, 3)                                   # MB / ms = GB/s
        groups[name] = g
    return groups


def training_step(vad, dev, hw, clips=32, t=10, steps=3, warmup=1):
    """frames/s through VideoTrainer.step (forward in train mode + MSE + backward + Adam as HIP kernels) on synthetic clips,
    default VideoAutoencoder(latent 128, hidden 128, 2 layers); FLOPs counted as 3 x the forward's.
    Each precision also gets a `roofline` object: the MFMA groups of the step (convolutions forward + data gradients, weight
    gradients) against the mode's dense matrix peak and the BatchNorm passes against HBM, from per-group hipEvents of ONE extra
    step (the events are recorded by the library around every launch group; the timed steps run without them)."""
    import numpy as np
    import torch
    lib = vad.hip.lib()
    m = vad.VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128, lstm_num_layers=2)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, 5).items()}, strict=True)
    m = m.to(dev)
    x = vad.scoring.synth_frames_device(0xC0FFEE + 4, 0, clips * t, hw, hw, 3, dev).view(clips, t, 3, hw, hw)
    out = {"unit": "frames/s trained", "clips": clips, "t": t, "dtype": "f32"}
    for precision in ("fp32", "split", "bf16", "winograd"):  # all start from the same weights: a fresh trainer re-reads the module
        state = {k: v.detach().clone() for k, v in m.state_dict().items()}
        tr = vad.VideoTrainer(m, precision=precision)
        first = None
        for _ in range(warmup):
            first = float(tr.step(x))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(x)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        res = {"value": round(clips * t / dt, 1), "ms_per_step": round(dt * 1e3, 3), "loss_first_last": [first, float(loss)],
               "algorithmic_tflops": round(3 * 3011510272.0 * (hw / 256.0) ** 2 * clips * t / dt / 1e12, 2)}
        # one more step with the per-group events on, in the SERIAL launch order (vad_debug_set_lstm_wavefront(0): no ConvLSTM layer
        # wavefront, weight gradients on the caller's stream) - beside each other the groups' event times overlap and say nothing
        # about a group; the timed steps above run the overlapped order
        vad.hip.check(lib.vad_prof_reset(), "vad_prof_reset")
        vad.hip.check(lib.vad_prof_enable(1), "vad_prof_enable")
        lib.vad_debug_set_lstm_wavefront(0)
        try:
            t1 = time.perf_counter()
            tr.forward_backward(x)
            torch.cuda.synchronize(dev)
            res["ms_per_step_serial_order"] = round((time.perf_counter() - t1) * 1e3, 3)
        finally:
            lib.vad_debug_set_lstm_wavefront(1)
        groups = training_groups(vad, clips * t, hw, precision)
        vad.hip.check(lib.vad_prof_enable(0), "vad_prof_enable")
        vad.hip.check(lib.vad_prof_reset(), "vad_prof_reset")
        mf = [g for n, g in groups.items() if n in TRAIN_GROUP_GFLOP and n not in ("first layer fwd", "first layer wgrad")]
        mf_ms = sum(g["ms"] for g in mf)
        mf_tf = sum(g["tflops"] * g["ms"] for g in mf) / mf_ms if mf_ms > 0 else 0.0
        bn = [g for n, g in groups.items() if n in TRAIN_GROUP_MB_FP32]
        bn_ms = sum(g["ms"] for g in bn)
        peak = TRAIN_PEAK_TFLOPS[precision]
        res["roofline"] = {"bound": "mfma", "kernel": "3x3 / transposed / 1x1 convolutions (forward + data gradients) and weight-gradient GEMMs behind the first layer",
                           "achieved": round(mf_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(mf_tf / peak, 4),
                           "ms": round(mf_ms, 3), "traffic": None,
                           "batchnorm_passes": {"bound": "hbm", "ms": round(bn_ms, 3), "share_of_step": round(bn_ms / res["ms_per_step_serial_order"], 3),
                                                "share_of": "the serial-order step the groups were timed in",
                                                "achieved": round(sum(g["algorithmic_GB"] for g in bn) / bn_ms, 3) if bn_ms > 0 else None,
                                                "peak": 8.0, "unit": "TB/s"},
                           "groups": groups}
        if precision == "fp32":
            out.update(res)
            out["workspace_GiB"] = round(tr._ws.numel() / 2**30, 2)
        elif precision == "split":
            out["split_precision"] = dict(res, arithmetic="3x3 / transposed convolutions (forward + data gradients) on split-fp16 operands, "
                                                          "everything else fp32")
        elif precision == "winograd":
            # (the group rates are ALGORITHMIC, direct-convolution FLOPs: the Winograd groups execute 16/36 of them, so their
            # `tflops` may exceed the 157.3 peak and `roofline.frac` is not an executed-FLOP fraction in this mode)
            res["roofline"]["flops"] = "algorithmic (direct-convolution) FLOPs; the 3x3 forward / data-gradient groups execute 16/36 of them"
            out["winograd_precision"] = dict(res, arithmetic="fp32 everywhere; the 3x3 convolutions behind the first layer (forward + data gradients, ConvLSTM gate "
                                                             "convolutions included) as Winograd F(2x2,3x3) on the exact-fp32 MFMA; weight gradients direct")
        else:
            out["bf16_precision"] = dict(res, arithmetic="BASELINE configs[4] dtype: activation and activation-gradient tensors bf16 in HBM, every convolution / "
                                                         "weight-gradient GEMM behind the first layer and the first layer's forward on bf16 MFMA operands with fp32 accumulation; arithmetic inside "
                                                         "the kernels, BatchNorm statistics, cell states, master weights, parameter gradients, loss, Adam fp32")
        m.load_state_dict(state)
        del tr
    del m, x
    torch.cuda.empty_cache()
    return out


def multi_gpu_evidence(vad, dist, args, state, scores, seed, hw, per_gpu, width, rank, world, dev, local_rank):
    """N > 1 makes the run self-verifying: which device every rank really used, the wall time of the one collective, and
    rank 0's parity check of the GATHERED vector against the CPU oracle on one item from every rank's block (so a rank that
    scored the wrong block, or a gather in the wrong order, fails here and not silently)."""
    from oracle import torch_oracle
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": torch.cuda.get_device_name(dev),
            "uuid": str(getattr(props, "uuid", "")), "cus": props.multi_processor_count, "hbm_GiB": round(props.total_memory / 2**30, 1)}
    ranks = [None] * world
    dist.all_gather_object(ranks, mine)
    # the collective alone: `per * width` floats per rank, as in every step
    local = torch.zeros(per_gpu, width, dtype=torch.float32, device=dev)
    gathered = torch.empty(world * per_gpu, width, dtype=torch.float32, device=dev)
    for _ in range(3):
        dist.all_gather_into_tensor(gathered, local)
    torch.cuda.synchronize()
    dist.barrier()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_gather_into_tensor(gathered, local)
    torch.cuda.synchronize()
    ag_ms = (time.perf_counter() - t0) / reps * 1e3
    tt = torch.tensor([ag_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ev = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks,
          "distinct_devices": len({(r["uuid"], r["device_index"]) for r in ranks}),
          "allgather_ms": round(float(tt.item()), 4), "allgather_bytes_per_rank": per_gpu * width * 4}
    if rank == 0 and args.workload != "dense" and not args.no_cpu_baseline:
        torch.set_num_threads(host_cores())
        items = [r * per_gpu + (37 * r + 5) % per_gpu for r in range(world)]       # one item out of every rank's block
        with torch.no_grad():
            if args.workload == "image":
                xs = torch.from_numpy(np.concatenate([vad.synth.frames(seed, i, 1, 3, hw, hw) for i in items]))
                ref = torch_oracle.img_scores(state, xs)["scores"]
            else:
                t = args.clip_len
                xs = torch.from_numpy(np.concatenate([vad.synth.clips(seed, i, 1, t, 3, hw, hw) for i in items]))
                ref = torch_oracle.vid_scores(state, xs, 128, 2)["frame"]
        got = scores[items].cpu()
        rel = float(((got - ref).abs() / ref.abs()).max())
        ev["parity"] = {"checked_items": items, "oracle": "oracle/torch_oracle.py on rank 0's host cores",
                        "max_rel_score_err_vs_cpu": rel, "within_1e-4": bool(rel < 1e-4)}
        if not rel < 1e-4:
            raise SystemExit(f"multi-GPU parity check failed: gathered scores differ from the CPU oracle by {rel:.3e} on items {items}")
    return ev


def host_cores() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a 1-GPU box exposes
    256 logical CPUs but grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(vad, state, kind, clip_len, gpu_scores, seed, hw):
    """The CPU restatement (oracle/torch_oracle.py — the reference itself cannot travel) on the host cores, on a
    bounded sample of the same workload (1 warm-up + 2 timed passes), plus the GPU-vs-CPU score
    parity on that sample."""
    from oracle import torch_oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    passes = 2
    with torch.no_grad():
        if kind == "image":
            n = min(64, int(gpu_scores.shape[0]))            # configs[0]: 64 frames, batches of 16 (evaluate.py:240); fewer when the step has fewer
            bs = min(16, n)
            xs = torch.from_numpy(vad.synth.frames(seed, 0, n, 3, hw, hw))
            torch_oracle.img_scores(state, xs[:bs])          # warm-up
            t0 = time.perf_counter()
            for _ in range(passes):
                ref = torch.cat([torch_oracle.img_scores(state, xs[i:i + bs])["scores"] for i in range(0, n, bs)])
            dt = (time.perf_counter() - t0) / passes
            got = gpu_scores[:n].cpu()
            frames, sample = n, f"{n} of the step's frames, batches of {bs}, 1 warm-up batch + {passes} timed passes (mean)"
        else:
            nclips, t = min(4, int(gpu_scores.shape[0])), clip_len
            xs = torch.from_numpy(vad.synth.clips(seed, 0, nclips, t, 3, hw, hw))
            torch_oracle.vid_scores(state, xs[:1], 128, 2)
            t0 = time.perf_counter()
            for _ in range(passes):
                ref = torch_oracle.vid_scores(state, xs, 128, 2)["frame"]
            dt = (time.perf_counter() - t0) / passes
            got = gpu_scores[:nclips].cpu()
            frames, sample = nclips * t, f"{nclips} clips x {t} frames in one batch, 1 warm-up clip + {passes} timed passes (mean)"
    rel = float(((got - ref).abs() / ref.abs()).max())
    return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample,
            "gpu_vs_cpu_max_rel_score_err": rel}


def reference_call_sizes(vad, lib, img_model, dev, hw, seed, steps=100, warmup=10, precision="fp32"):
    """Latency / throughput of the batch sizes the reference's own scripts use: inputs resident in HBM,
    `steps` calls back to back after `warmup`, one synchronise at the end."""
    def timed(fn):
        with torch.no_grad():
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps

    lib.vad_prof_enable(0)
    out = {}
    was = img_model.precision
    img_model.precision = precision
    frames = vad.scoring.synth_frames_device(seed + 5, 0, 16, hw, hw, device=dev)
    for b, site in ((1, "main.py:274"), (16, "evaluate.py:240")):
        x = frames[:b].contiguous()
        dt = timed(lambda: img_model.get_reconstruction_error(x))
        out[f"image_batch_{b}"] = {"ms": round(dt * 1e3, 4), "frames_per_sec": round(b / dt, 1), "call_site": site}
    vm = vad.VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128, lstm_num_layers=2)
    shapes = {k: tuple(v.shape) for k, v in vm.state_dict().items()}
    vm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, 8).items()}, strict=True)
    vm = vm.to(dev).eval()
    vm.precision = precision
    img_model.precision = was
    clips = vad.scoring.synth_frames_device(seed + 6, 0, 4 * 16, hw, hw, device=dev).view(4, 16, 3, hw, hw)
    for b, site in ((4, "evaluate_video.py:416"), (1, "evaluate_video.py:344 (one window)")):
        x = clips[:b].contiguous()
        dt = timed(lambda: vm.get_reconstruction_error(x, per_frame=True))
        out[f"video_{b}x16"] = {"ms": round(dt * 1e3, 4), "frames_per_sec": round(b * 16 / dt, 1), "call_site": site}
    del vm, clips, frames
    torch.cuda.empty_cache()
    return out


def video_config2(vad, hip, lib, dev, hw, steps, warmup, with_cpu, clips=64, t=10):
    """BASELINE configs[2] beside the headline in the default line: ConvLSTM video autoencoder scoring, 64 clips x 10 frames
    resident in HBM, same step / warm-up counts, with its own `roofline` (dominant kernel: the conv3x3 MFMA kernel in its
    encoder and ConvLSTM forms) and `cpu_baseline`.  Never part of `value`."""
    seed = 0xC0FFEE + 2
    model = vad.VideoAutoencoder(in_channels=3, latent_dim=128, lstm_hidden_dim=128, lstm_num_layers=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in vad.synth.synthetic_state(shapes, 8).items()}
    model.load_state_dict(state, strict=True)
    model = model.to(dev).eval()
    x = vad.scoring.synth_frames_device(seed, 0, clips * t, hw, hw, device=dev).view(clips, t, 3, hw, hw)
    with torch.no_grad():
        for _ in range(warmup):
            scores = model.get_reconstruction_error(x, per_frame=True)
        torch.cuda.synchronize(dev)
        lib.vad_prof_reset()
        lib.vad_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            scores = model.get_reconstruction_error(x, per_frame=True)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        lib.vad_prof_enable(0)
    fps = clips * t * steps / elapsed
    scale = (hw * hw) / 65536.0
    layers, roofline = layers_and_roofline(hip, lib, "video", hw, clips, steps, t, 1, hw == 256 and int(model.chunk) == 64, fps,
                                           VID_FLOP_PER_FRAME * scale, VID_BYTES_PER_FRAME * scale)
    out = {"value": round(fps, 1), "unit": "frames/s", "clips_per_sec": round(fps / t, 1), "ms_per_step": round(elapsed / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "dtype": "f32",
           "config": {"workload": f"configs[2]: ConvLSTM video autoencoder scoring, {clips} clips x {t} frames of {hw}x{hw}x3 per GPU",
                      "frames_per_gpu_per_step": clips * t, "chunk": int(model.chunk)},
           "roofline": roofline, "layers": layers}
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(vad, state, "video", t, scores, seed, hw)
    # the opt-in Winograd mode on the same clips (the encoder's 3x3 convolutions; the ConvLSTM cell stays direct)
    exact = scores.clone()
    model.precision = "winograd"
    with torch.no_grad():
        for _ in range(max(1, warmup)):
            scores = model.get_reconstruction_error(x, per_frame=True)
        torch.cuda.synchronize(dev)
        lib.vad_prof_reset()
        lib.vad_prof_enable(1)
        t1 = time.perf_counter()
        for _ in range(steps):
            scores = model.get_reconstruction_error(x, per_frame=True)
        torch.cuda.synchronize(dev)
        el2 = time.perf_counter() - t1
        lib.vad_prof_enable(0)
    out["winograd_precision"] = dict({"value": round(clips * t * steps / el2, 1), "unit": "frames/s", "ms_per_step": round(el2 / steps * 1e3, 3),
                                      "max_rel_score_diff_vs_exact_fp32": float(((scores - exact).abs() / exact.abs()).max()),
                                      "speedup_vs_direct": round(elapsed / el2, 3)}, **winograd_roofline(hip, lib, "video", hw, clips, steps, t))
    del model, x
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
