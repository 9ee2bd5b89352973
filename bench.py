"""Training clips/sec of the hot path on N MI355X (one process per GPU, RCCL over xGMI).

    python bench.py [--gpus N --steps K --warmup W] [--hw 224|40] [--batch 32] [--length 32]

A step = one optimisation step of the tri-modal LFAN on one resident synthetic batch per
rank: zero_grad -> IR-50 forward over B*L frames (frozen, fp32 MFMA) -> TCNs -> cross-modal
fusion -> regressor -> cross-entropy -> backward through the trainable tail -> gradient
all-reduce (N > 1) -> Nesterov SGD (reference trainer.py:345-391).  Prints ONE JSON line on
rank 0.  `roofline` is measured live with HIP events around the IR-50 forward (all
cer::conv_igemm_kernel launches, >99 % of the step's FLOPs); `cpu_baseline` times the CPU
oracle on a bounded sample of the same workload on this box's host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.modules.setdefault("triton", None)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak; the bf16x3 kernels spend 3 MFMAs per product -> 833.3 effective
MODS = ["video", "vggish", "bert"]
# kernel variant ids reported by cer_conv2d_b3_tile -> names as rocprofv3 prints them (csrc/conv_b3.hip)
B3_KERNEL_NAMES = {41: "cer::conv_b3_dma16_kernel<128, 128, 2, 2, 4, 2>", 42: "cer::conv_b3_dma16_kernel<128, 64, 2, 2, 4, 2>",
                   44: "cer::conv_b3_dma16_kernel<64, 128, 1, 4, 4, 2>", 45: "cer::conv_b3_dma16_kernel<64, 64, 2, 2, 4, 2>",
                   46: "cer::conv_b3_dma16_kernel<256, 256, 2, 4, 8, 2>", 48: "cer::conv_b3_dma16_kernel<256, 64, 4, 1, 4, 2>"}


def ir50_forward_flops(hw):
    """Algorithmic FLOPs (2/MAC) of one IR-50 frame: 52 convs + head FC."""
    from feature_vs_text_compound_emotion_amd.synth import ir50_units
    h = hw
    macs = h * h * 64 * 3 * 9
    for cin, depth, stride in ir50_units():
        macs += h * h * depth * cin * 9            # conv1 3x3 s1 at input resolution
        ho = (h - 1) // stride + 1
        macs += ho * ho * depth * depth * 9        # conv2 3x3 stride s
        if cin != depth:
            macs += ho * ho * depth * cin          # 1x1 projection shortcut
        h = ho
    macs += 512 * h * h * 512                      # head FC
    return 2.0 * macs


def build_model(hw, length, device):
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import LFAN
    sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=hw // 8, seed=0)
    model = LFAN(backbone_settings={}, output_dim=7, task="CLASSIFICATION", modality=MODS, example_length=length,
                 kernel_size=5, tcn_channel=synth.TCN_CHANNELS, root_dir="", device=device, head_hw=hw // 8)
    model.init(load_backbone=False)
    model.load_state_dict(sd, strict=True)
    return model.to(device), sd


def measured_traffic(hw, batch, length, encoders, precision, kernel=None):
    """HBM bytes per step of cer::conv_igemm_kernel from the committed rocprofv3 PMC passes
    (tools/collect_traffic.py: FETCH_SIZE x2 + WRITE_SIZE, separate passes, gfx950 corrections).  PMC
    counters cannot be read from inside the timed process, so the value is looked up by configuration
    and is null when no profile of this exact configuration has been committed."""
    path = os.path.join(ROOT, "profiles", f"round1b_traffic_{precision}_hw{hw}.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None
    if (str(t.get("batch")), str(t.get("length")), t.get("encoders")) != (str(batch), str(length), encoders):
        return None
    if kernel is not None:  # per launch of one kernel variant, like roofline.achieved
        return t.get("per_kernel", {}).get(kernel, {}).get("hbm_bytes_per_launch")
    return t["hbm_bytes_per_step"]


def host_threads():
    """Threads for the CPU baseline.  The GPU box gives a job a CFS quota (cpu.max) far below the
    visible CPU count; torch's default of one thread per visible CPU then gets throttled (measured on
    the round-1 box, 16-CPU quota of a 2 x EPYC 9575F: 16 threads 0.45, 32 threads 0.54, 128 threads
    0.18 TFLOP/s sustained on a 3x3 conv), so use twice the quota, capped by the visible CPUs."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, 2 * int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return n


def build_extractor(batch, rank, dev):
    """VGGish + BERT with seeded weights and one resident raw batch: 1 s of 16 kHz PCM and a 64-slot
    sentence (63 tokens + 1 pad: the reference's exclude_padding rejects a sentence with no pad)."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.audio_backbone import AudioBackbone
    from feature_vs_text_compound_emotion_amd.feature_extractor import MultimodalFeatureExtractor
    from feature_vs_text_compound_emotion_amd.text_encoder import BertEncoderHIP
    ab, te = AudioBackbone(), BertEncoderHIP()
    ab.backbone.load_state_dict(synth.make_state_dict(synth.vggish_spec(""), seed=21), strict=True)
    te.load_state_dict(synth.make_state_dict(synth.bert_spec(""), seed=31), strict=True)
    fx = MultimodalFeatureExtractor(ab, te, fps=32).to(dev).eval()
    pcm = torch.stack([synth.make_audio_int16(1.0, 16000, seed=5000 + rank * 1000 + i) for i in range(batch)])
    ids, mask = synth.make_token_ids(batch, 64, seed=900 + rank, pad_from=[63] * batch)
    return fx, {"pcm": pcm.to(dev), "ids": ids.to(dev), "mask": mask.to(dev), "mask_cpu": mask}


def cpu_baseline(hw, length, encoders=True):
    """CPU oracle, same step, bounded sample (about 10-30 s)."""
    from feature_vs_text_compound_emotion_amd import synth
    from oracle.lfan import cross_entropy_mean, lfan_forward, sgd_nesterov_step
    threads = host_threads()
    torch.set_num_threads(threads)
    sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=hw // 8, seed=0)
    alias = synth.lfan_spec(MODS, head_hw=hw // 8)[1]
    names = [k for k in sd if not k.startswith("spatial.") and k not in alias
             and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    clips, steps = (8, 3) if hw <= 64 else (1, 1)  # ~10-15 s of CPU work either way

    if encoders:
        import oracle
        from feature_vs_text_compound_emotion_amd.feature_extractor import align_tokens_to_frames
        vsd = synth.make_state_dict(synth.vggish_spec(""), seed=21)
        bsd = synth.make_state_dict(synth.bert_spec(""), seed=31)

    def encode(xs, b, l):
        """oracle VGGish + BERT on raw audio / token ids (same work as the GPU step)."""
        aud, txt = [], []
        ids, mask = synth.make_token_ids(b, 64, seed=900, pad_from=[63] * b)
        with torch.no_grad():
            tok = oracle.bert_token_features(ids, mask, bsd)
            for i in range(b):
                ex = oracle.wav_int16_to_examples(synth.make_audio_int16(1.0, 16000, seed=5000 + i).numpy(), 16000, 0.96,
                                                  1.0 / 32)[:l]
                aud.append(oracle.vggish_forward(ex.astype("float32"), vsd))
                words = oracle.exclude_padding(tok[i:i + 1], mask[i:i + 1])
                txt.append(words[torch.tensor(align_tokens_to_frames(words.shape[0], l))])
        xs["vggish"] = torch.stack(aud).unsqueeze(1)
        xs["bert"] = torch.stack(txt).unsqueeze(1)
        return xs

    def step(b, l, h, sdx, enc=False):
        xs, ls = synth.make_clip_batch(MODS, b, l, hw=h, seed=7)
        if enc:
            xs = encode(xs, b, l)
        params = [sdx[n].clone().requires_grad_(True) for n in names]
        s = dict(sdx)
        s.update(zip(names, params))
        for a, src in alias.items():
            s[a] = s[src]
        loss = cross_entropy_mean(lfan_forward(xs, s, MODS, train=True, backbone_train=True), ls)
        grads = torch.autograd.grad(loss, params)
        sgd_nesterov_step([p.detach() for p in params], list(grads), [None] * len(params))
        return loss.item()

    warm_sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=5, seed=0) if hw != 40 else sd
    step(1, length, 40, warm_sd)  # thread-pool / allocator warm-up, untimed
    t0 = time.perf_counter()
    for _ in range(steps):
        step(clips, length, hw, sd, enc=encoders)
    dt = time.perf_counter() - t0
    return {"value": clips * steps / dt, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train step(s) of {clips} clip(s) x {length} frames x {hw}x{hw} (tri-modal LFAN"
                      f"{' incl. VGGish + BERT encoders' if encoders else ''}, oracle on torch-CPU fp32, "
                      f"{threads} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hw", type=int, default=224, help="frame size: 224 (north-star shape) or 40 (reference crop)")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--length", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra 3 steps on the fp32 kernels (N = 1 only)")
    ap.add_argument("--precision", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="IR-50 conv kernels: bf16x3 = split hi/lo bf16 operands, 3 bf16 MFMAs per product, fp32-class "
                         "accuracy (logit error ~1e-6); fp32 = exact fp32 MFMA")
    ap.add_argument("--release", type=int, default=0, choices=[0, 1, 2, 3],
                    help="gradual-release groups of the reference's ResnetParamControl to un-freeze before timing "
                         "(0 = as the reference trains: encoder frozen; 1 = output layer; 2 = + stage 4; 3 = + half of stage 3)")
    ap.add_argument("--encoders", choices=["on", "off"], default="on",
                    help="on: VGGish (log-mel from 1 s PCM) and BERT (64 tokens) run on the GPU inside the step; "
                         "off: pre-computed per-frame features, as the reference trainer feeds them")
    a = ap.parse_args()

    from feature_vs_text_compound_emotion_amd import ops, synth
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD, init_process_group_from_env
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss

    # CER_BENCH_BACKEND=gloo rehearses the N>1 path on a one-GPU box (all ranks share cuda:0); the driver's
    # real multi-GPU runs use the default: RCCL ("nccl"), one rank per GPU
    rank, world, local = init_process_group_from_env(os.environ.get("CER_BENCH_BACKEND"))
    if os.environ.get("CER_BENCH_BACKEND") == "gloo" and torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model, _ = build_model(a.hw, a.length, dev)
    model.spatial["visual"].backbone.precision = a.precision
    model.train()
    if a.release:
        from feature_vs_text_compound_emotion_amd.parameter_control import ResnetParamControl
        pc = ResnetParamControl(trainer=None, release_count=a.release)
        for _ in range(a.release):
            pc.release_param(model.spatial)
    ddp = ClipDataParallel(model, world_size=world)
    # the reference's torch.optim.SGD(momentum .9, nesterov, wd 1e-4, lr 1e-3 -- F8) as one fused launch over flat buffers
    opt = FlatNesterovSGD(ddp, lr=1e-3, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=True)
    x, labels = synth.make_clip_batch(MODS, a.batch, a.length, hw=a.hw, seed=1234 + rank)
    x = {k: v.to(dev) for k, v in x.items()}
    labels = labels.to(dev)
    fx = None
    if a.encoders == "on":
        fx, raw = build_extractor(a.batch, rank, dev)

    ev = []

    def pre(mod, inp):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append([e, None])

    def post(mod, inp, out):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[-1][1] = e

    model.spatial["visual"].register_forward_pre_hook(pre)
    model.spatial["visual"].register_forward_hook(post)

    def step():
        ddp.zero_grad()
        inputs = x
        if fx is not None:  # raw audio + token ids -> per-frame features on the GPU (frozen encoders)
            inputs = fx(x["video"], raw["pcm"], raw["ids"], raw["mask"], raw["mask_cpu"])
        out = model(inputs)
        loss = cross_entropy_loss(out, labels)
        loss.backward()
        ddp.all_reduce_gradients()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    ev.clear()
    trace = ops.CONV_TRACE = [] if a.precision == "bf16x3" else None  # HIP events around every bf16x3 conv launch
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ops.CONV_TRACE = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    enc_ms = sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
    frames = a.batch * a.length
    flops = ir50_forward_flops(a.hw) * frames
    achieved = flops / (enc_ms * 1e-3) / 1e12
    b3 = a.precision == "bf16x3"
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if b3 else FP32_MFMA_PEAK_TFLOPS

    # per kernel variant: algorithmic FLOPs of its launches / their HIP-event durations (launch stream), over the timed steps
    kernels = {}
    for tile, fl, e0, e1, nbytes in (trace or []):
        k = kernels.setdefault(B3_KERNEL_NAMES.get(tile, f"conv_b3 tile {tile}"),
                               {"launches": 0, "flops": 0.0, "ms": 0.0, "algo_bytes": 0.0})
        k["launches"] += 1
        k["flops"] += fl
        k["algo_bytes"] += nbytes
        k["ms"] += e0.elapsed_time(e1)
    for k in kernels.values():
        k["launches_per_step"] = k["launches"] / a.steps
        k["avg_launch_ms"] = k["ms"] / k["launches"]
        k["flops_per_launch"] = k["flops"] / k["launches"]
        k["achieved_tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12
        k["frac"] = k["achieved_tflops"] / peak
        k["ms_per_step"] = k.pop("ms") / a.steps
        k["algorithmic_bytes_per_launch"] = k["algo_bytes"] / k["launches"]
        del k["flops"]
    dominant = max(kernels, key=lambda n: kernels[n]["ms_per_step"]) if kernels else None

    alt = None
    if world == 1 and b3 and not a.no_alt:
        # the same step on the exact-fp32 MFMA kernels (2 timed steps): the other point of the
        # accuracy/throughput trade-off, priced against ITS roofline (fp32 MFMA peak)
        model.spatial["visual"].backbone.precision = "fp32"
        step()
        ev.clear()
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        dt1 = (time.perf_counter() - t1) / 2
        ms1 = sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
        ach1 = flops / (ms1 * 1e-3) / 1e12
        alt = {"conv_precision": "fp32", "value": a.batch / dt1, "unit": "clips/s", "ms_per_step": dt1 * 1e3,
               "roofline": {"bound": "mfma", "kernel": "cer::conv_igemm_kernel (v_mfma_f32_32x32x2_f32)", "achieved": ach1,
                            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach1 / FP32_MFMA_PEAK_TFLOPS,
                            "ms_per_step_in_kernel": ms1}}
        model.spatial["visual"].backbone.precision = a.precision
    span = {"what": ("whole IR-50 forward (51 bf16x3 implicit-GEMM convs + head FC, fp32 stem, batch-statistics BatchNorm "
                     "passes): algorithmic IR-50 FLOPs of the step / HIP-event span of the encoder forward" if b3 else
                     "whole IR-50 forward on cer::conv_igemm_kernel (v_mfma_f32_32x32x2_f32)"),
            "achieved": achieved, "frac": achieved / peak, "ms_per_step_in_kernel": enc_ms, "algorithmic_flops_per_step": flops}
    traffic_note = ("HBM bytes per step over all conv kernel launches (rocprofv3 PMC passes, profiles/*traffic*_hw*.json: "
                    "FETCH_SIZE x2 + WRITE_SIZE; the x2 read correction is calibrated for 128-B requests and may over-count "
                    "64-B row segments); algorithmic minimum 18.6 MB/frame @40x40, 584 MB/frame @224x224 (SURVEY 8d)")
    if dominant is not None:
        kd = kernels[dominant]
        roofline = {"bound": "mfma", "kernel": dominant, "achieved": kd["achieved_tflops"], "peak": peak, "unit": "TFLOP/s",
                    "frac": kd["frac"], "peak_note": "dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product (bf16x3)",
                    "avg_launch_ms": kd["avg_launch_ms"], "launches_per_step": kd["launches_per_step"],
                    "algorithmic_flops_per_launch": kd["flops_per_launch"],
                    "definition": "algorithmic FLOPs (2*M*Cout*Cin*KH*KW) of this kernel's launches / their summed HIP-event "
                                  "durations on the launch stream over the timed steps",
                    "traffic": measured_traffic(a.hw, a.batch, a.length, a.encoders, a.precision, dominant),
                    "traffic_note": "HBM bytes per launch of this kernel (rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes, "
                                    "FETCH x2 per the gfx950 correction, profiles/round1b_traffic_*.json); algorithmic bytes per "
                                    "launch = split input + output tensors + weights",
                    "algorithmic_bytes_per_launch": kd["algo_bytes"] / kd["launches"],
                    "traffic_per_step_all_convs": measured_traffic(a.hw, a.batch, a.length, a.encoders, a.precision),
                    "all_kernels": kernels, "encoder_span": span}
    else:
        roofline = {"bound": "mfma", "kernel": "cer::conv_igemm_kernel (IR-50 forward: 52 implicit-GEMM convs + head FC, "
                                               "v_mfma_f32_32x32x2_f32)", "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": "fp32 MFMA peak",
                    "traffic": measured_traffic(a.hw, a.batch, a.length, a.encoders, a.precision), "traffic_note": traffic_note,
                    "algorithmic_flops_per_step": flops, "ms_per_step_in_kernel": enc_ms}
    if rank == 0:
        res = {
            "metric": "training clips/sec (32-frame tri-modal clip)",
            "value": world * a.batch * a.steps / dt,
            "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (split hi/lo bf16 operands, fp32 accumulate; fp32-class accuracy)" if b3 else "f32",
            "data": "synthetic",
            "config": {"workload": f"LFAN tri-modal training step: frozen IR-50 forward on {a.batch}x{a.length} frames "
                                   f"of {a.hw}x{a.hw} + TCN/fusion/regressor forward+backward + CE + fused Nesterov SGD; "
                                   + ("VGGish (log-mel of 1 s PCM, 32 examples/clip) and BERT-base (64-token sentence) run on "
                                      "the GPU inside the step" if a.encoders == "on" else
                                      "vggish/bert as pre-computed per-frame features (as the reference trainer feeds them)"),
                       "encoders_on_gpu": a.encoders == "on",
                       "clips_per_gpu": a.batch, "global_batch": a.batch * world, "frames_per_clip": a.length,
                       "frame_hw": a.hw, "n_classes": 7, "released_encoder_groups": a.release,
                       "trainable_parameters": int(sum(p.numel() for p in ddp.params)), "conv_precision": a.precision, "parallelism": f"dp{world} over clips, flat-bucket RCCL all-reduce",
                       "loss": float(loss.item())},
            "roofline": roofline,
        }
        if alt is not None:
            res["fp32_path"] = alt
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.hw, a.length, a.encoders == "on")
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
