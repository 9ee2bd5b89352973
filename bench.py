"""Training clips/sec of the hot path on N MI355X (one process per GPU, RCCL over xGMI).

    python bench.py [--gpus N --steps K --warmup W] [--hw 224|40] [--batch 32] [--length 32]
                    [--precision bf16x3|fp32|bf16|fp16] [--model LFAN|JMT|MT|CAN] [--n-cls 7]

A step = one optimisation step on one resident synthetic batch per rank: zero_grad -> (VGGish + BERT encoders on raw
audio / token ids) -> frozen IR-50 forward over B*L frames in the reference's model.train() mode -> TCNs -> fusion head ->
classifier -> cross-entropy -> backward through the trainable tail -> gradient all-reduce (N > 1) -> Nesterov SGD
(reference trainer.py:345-391).  Prints ONE JSON line on rank 0.

`roofline` is measured live: HIP events on the launch stream around every conv launch of the vision encoder
(ops.CONV_TRACE), per kernel variant: algorithmic FLOPs / summed durations.  Peak: dense bf16/f16 MFMA 2500 TFLOP/s for the
narrow kernels (precision bf16 / fp16, one MFMA per product), 2500 / 3 for bf16x3 (three MFMAs per product), 157.3 for fp32.
At N = 1 the default run adds, inside `roofline.other_configs`, the same workload on the other precisions (fp32 with
>= 5 timed steps, fp16 and bf16 narrow storage), BASELINE cfg5 (64-frame clips, 8 classes, bf16 storage) and cfg3
(video + vggish, JMT head: the MFMA attention at 1024 tokens x 6 stacks, with its own roofline), and `cpu_baseline`
times the CPU oracle on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import hashlib
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak (same rate); bf16x3 spends 3 MFMAs per product -> 833.3 effective
ALL_MODS = ["video", "vggish", "bert"]
NARROW = ("bf16", "fp16")
# kernel variant ids reported by cer_conv2d_b3_tile / cer_conv2d_n16_tile -> names as rocprofv3 prints them
KERNEL_NAMES = {41: "cer::conv_b3_dma16_kernel<128, 128, 2, 2, 4, 2>", 42: "cer::conv_b3_dma16_kernel<128, 64, 2, 2, 4, 2>",
                44: "cer::conv_b3_dma16_kernel<64, 128, 1, 4, 4, 2>", 45: "cer::conv_b3_dma16_kernel<64, 64, 2, 2, 4, 2>",
                48: "cer::conv_b3_dma16_kernel<256, 64, 4, 1, 4, 2>",
                51: "cer::conv_b3_s2d_kernel<64>", 52: "cer::conv_b3_s2d_kernel<128>",
                53: "cer::conv_b3_win_kernel<64, 4, 1, true>", 56: "cer::conv_b3_win_kernel<128, 4, 2, false>",
                58: "cer::conv_b3_patch_kernel<128, 4, 2, false>", 59: "cer::conv_b3_patch_kernel<64, 4, 1, true>",
                71: "cer::conv_n16_patch_kernel<64, 4, 1, 1, {f16}, false>", 72: "cer::conv_n16_patch_kernel<128, 4, 2, 2, {f16}, false>",
                78: "cer::conv_n16_patch_kernel<128, 4, 2, 2, {f16}, true>",
                73: "cer::conv_n16_win_kernel<64, 4, 2, {f16}, false, 2>", 76: "cer::conv_n16_win_kernel<128, 4, 2, {f16}, true, 2>",
                77: "cer::conv_n16_win_kernel<64, 4, 1, {f16}, false, 1>", 79: "cer::conv_n16_p64_kernel<{f16}>",
                81: "cer::conv_n16_s2d_kernel<64, {f16}>", 82: "cer::conv_n16_s2d_kernel<128, {f16}>",
                91: "cer::conv_n16_kernel<256, 256, 2, 4, {f16}, 2>", 94: "cer::conv_n16_kernel<128, 128, 2, 2, {f16}, 2>",
                63: "cer::conv_n16_kernel<256, 64, 4, 1, {f16}, 1>", 64: "cer::conv_n16_kernel<128, 128, 2, 2, {f16}, 1>",
                65: "cer::conv_n16_kernel<128, 64, 2, 2, {f16}, 1>", 66: "cer::conv_n16_kernel<64, 64, 2, 2, {f16}, 1>",
                67: "cer::conv_n16_kernel<64, 128, 1, 4, {f16}, 1>"}


def peak_tflops(precision):
    if precision == "fp32":
        return FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"
    if precision == "bf16x3":
        return BF16_MFMA_PEAK_TFLOPS / 3.0, "dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product (bf16x3)"
    return BF16_MFMA_PEAK_TFLOPS, "dense bf16 / f16 MFMA peak, one MFMA per product (narrow storage)"


def dtype_string(precision):
    return {"bf16x3": "bf16x3 (split hi/lo bf16 operands, 3 MFMAs per product, fp32 accumulate; <= 2^-15 relative per product)",
            "fp32": "f32", "bf16": "bf16 (one 16-bit plane per tensor, one MFMA per product, fp32 accumulate and epilogue)",
            "fp16": "f16 (one 16-bit plane per tensor, one MFMA per product, fp32 accumulate and epilogue)"}[precision]


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


def kernel_source_sha():
    """Identity of the code the traffic profiles were taken on: the conv / BatchNorm kernel sources."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "feature_vs_text_compound_emotion_amd", "csrc")
    for f in ("conv_common.h", "conv_b3.hip", "conv_b3_patch.hip", "conv_b3_s2d.hip", "conv_n16.hip", "conv_n16_s2d.hip", "conv_n16_patch.hip", "conv_n16_p64.hip", "conv_igemm.hip", "stem_conv.hip", "encoder_bn.hip"):
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(cfg, kernel=None):
    """HBM bytes per launch (or per step over all conv launches) from the committed rocprofv3 PMC passes
    (tools/collect_traffic.py: FETCH_SIZE x2 + WRITE_SIZE, separate passes, gfx950 corrections).  PMC counters cannot be read
    from inside the timed process, so the value is looked up -- by configuration AND by the hash of the kernel sources the
    profile was taken on: a profile of other code is not reported (null)."""
    t = None
    for rnd in ("round3", "round2"):     # the newest profile taken on THESE kernel sources
        path = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{cfg['precision']}_hw{cfg['hw']}_L{cfg['length']}.json")
        try:
            c = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (str(c.get("batch")), str(c.get("length")), c.get("encoders"), c.get("kernel_source_sha")) == \
                (str(cfg["batch"]), str(cfg["length"]), cfg["encoders"], kernel_source_sha()):
            t = c
            break
    if t is None or cfg.get("release"):
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


def cpu_baseline(hw, length, encoders=True):
    """CPU oracle, same step, bounded sample (about 10-30 s)."""
    from feature_vs_text_compound_emotion_amd import synth
    from oracle.lfan import cross_entropy_mean, lfan_forward, sgd_nesterov_step
    MODS = ALL_MODS
    threads = host_threads()
    torch.set_num_threads(threads)
    sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=hw // 8, seed=0)
    alias = synth.lfan_spec(MODS, head_hw=hw // 8)[1]
    names = [k for k in sd if not k.startswith("spatial.") and k not in alias
             and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    clips, steps = (8, 3) if hw <= 64 else (1, 3)  # three timed steps; ~10-40 s of CPU work either way

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


class Workload:
    """One model + one resident synthetic batch per rank + the optimisation step around it."""

    def __init__(self, cfg, rank, world, dev):
        from feature_vs_text_compound_emotion_amd import synth
        from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD
        self.cfg, self.world, self.dev = cfg, world, dev
        mods, hw, length, n_cls = cfg["modalities"], cfg["hw"], cfg["length"], cfg["n_cls"]
        name = cfg["model"]
        if name == "LFAN":
            from feature_vs_text_compound_emotion_amd.lfan import LFAN
            sd = synth.lfan_state_dict(mods, n_cls=n_cls, head_hw=hw // 8, seed=0)
            model = LFAN(backbone_settings={}, output_dim=n_cls, task="CLASSIFICATION", modality=mods, example_length=length,
                         kernel_size=5, tcn_channel=synth.TCN_CHANNELS, root_dir="", device=dev, head_hw=hw // 8)
            model.init(load_backbone=False)
        else:
            from feature_vs_text_compound_emotion_amd.fusion_heads import CAN, JMT
            spec, alias = synth.can_spec(mods, n_cls, hw // 8) if name == "CAN" else synth.jmt_spec(mods, name, n_cls, hw // 8)
            sd = synth.make_state_dict(spec, alias, seed=0)
            kw = dict(task="CLASSIFICATION", modalities=mods, tcn_settings=synth.TCN_SETTINGS, backbone_settings={},
                      output_dim=n_cls, root_dir="", device=dev, head_hw=hw // 8, load_backbone=False)
            model = CAN(**kw) if name == "CAN" else JMT(model_name=name, **kw)
        model.load_state_dict(sd, strict=True)
        del sd
        self.model = model.to(dev)
        self.model.spatial["visual"].backbone.precision = cfg["precision"]
        self.model.train()
        if cfg["release"]:
            from feature_vs_text_compound_emotion_amd.parameter_control import ResnetParamControl
            pc = ResnetParamControl(trainer=None, release_count=min(cfg["release"], 3))
            for _ in range(min(cfg["release"], 3)):
                pc.release_param(self.model.spatial)
            if cfg["release"] == 4:   # extension (BASELINE configs[1]): backward through the whole IR-50, stem included
                bb = self.model.spatial["visual"].backbone
                for part in (bb.input_layer, bb.body, bb.output_layer):
                    for p in part.parameters():
                        p.requires_grad = True
            self.model.spatial["visual"].backbone.activation_memory = cfg["act_mem"]
        # released encoder units (28-43 M parameters): exchange 25 MB slices of the bucket under the backward of the units below
        self.ddp = ClipDataParallel(self.model, world_size=world, overlap=cfg["release"] > 0)
        # the reference's torch.optim.SGD(momentum .9, nesterov, wd 1e-4, lr 1e-3 -- F8) as one fused launch over flat buffers
        self.opt = FlatNesterovSGD(self.ddp, lr=1e-3, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=True)
        x, labels = synth.make_clip_batch(mods, cfg["batch"], length, hw=hw, seed=1234 + rank, n_cls=n_cls)
        self.x = {k: v.to(dev) for k, v in x.items()}
        self.labels = labels.to(dev)
        self.fx = self.raw = None
        if cfg["encoders"] == "on":
            self._build_extractor(rank)
        self.ev = []
        vis = self.model.spatial["visual"]
        self._hooks = [vis.register_forward_pre_hook(self._pre), vis.register_forward_hook(self._post)]

    def _build_extractor(self, rank):
        """VGGish + BERT with seeded weights and one resident raw batch: L/32 s of 16 kHz PCM (one 0.96 s example per
        video frame at 32 fps) and a 64-slot sentence (63 tokens + 1 pad: the reference's exclude_padding rejects a sentence
        with no pad)."""
        from feature_vs_text_compound_emotion_amd import synth
        from feature_vs_text_compound_emotion_amd.audio_backbone import AudioBackbone
        from feature_vs_text_compound_emotion_amd.feature_extractor import MultimodalFeatureExtractor
        from feature_vs_text_compound_emotion_amd.text_encoder import BertEncoderHIP
        cfg, dev = self.cfg, self.dev
        ab = AudioBackbone()
        ab.backbone.load_state_dict(synth.make_state_dict(synth.vggish_spec(""), seed=21), strict=True)
        if cfg["precision"] in NARROW:   # the narrow modes cover the whole tri-modal step: VGGish and BERT follow IR-50
            ab.backbone.precision = cfg["precision"]
        te = None
        if "bert" in cfg["modalities"]:
            te = BertEncoderHIP()
            te.load_state_dict(synth.make_state_dict(synth.bert_spec(""), seed=31), strict=True)
            if cfg["precision"] in NARROW or cfg["precision"] == "bf16x3":   # BERT's GEMMs in the encoder's arithmetic (fp32: exact)
                te.precision = cfg["precision"]
        self.fx = MultimodalFeatureExtractor(ab, te if te is not None else torch.nn.Identity(), fps=32).to(dev).eval()
        secs = cfg["length"] / 32.0
        pcm = torch.stack([synth.make_audio_int16(secs, 16000, seed=5000 + rank * 1000 + i) for i in range(cfg["batch"])])
        self.raw = {"pcm": pcm.to(dev)}
        if te is not None:
            ids, mask = synth.make_token_ids(cfg["batch"], 64, seed=900 + rank, pad_from=[63] * cfg["batch"])
            self.raw.update(ids=ids.to(dev), mask=mask.to(dev), mask_cpu=mask)

    def _pre(self, mod, inp):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append([e, None])

    def _post(self, mod, inp, out):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev[-1][1] = e

    def inputs(self):
        if self.fx is None:
            return dict(self.x)      # the model overwrites the dict it is handed, like the reference (model.py:487-515)
        length = self.cfg["length"]
        out = {}
        for m in self.cfg["modalities"]:  # raw audio + token ids -> per-frame features on the GPU (frozen encoders)
            if m == "video":
                out[m] = self.x["video"]
            elif m == "vggish":
                out[m] = self.fx.audio_features(self.raw["pcm"], length)
            else:
                out[m] = self.fx.text_features(self.raw["ids"], self.raw["mask"], length, self.raw["mask_cpu"])
        return out

    def step(self):
        from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
        self.ddp.zero_grad()
        out = self.model(self.inputs())
        loss = cross_entropy_loss(out, self.labels)
        loss.backward()
        self.ddp.all_reduce_gradients()
        self.opt.step()
        return loss

    def fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(self, steps, warmup):
        """W untimed + exactly K timed steps (barrier + synchronize on both sides, max over ranks) -> result dict."""
        from feature_vs_text_compound_emotion_amd import ops
        cfg = self.cfg
        for _ in range(warmup):
            self.step()
        self.ev.clear()
        trace = ops.CONV_TRACE = [] if cfg["precision"] != "fp32" else None  # HIP events around every encoder conv launch
        atrace = ops.ATTN_TRACE = [] if cfg["model"] in ("JMT", "MT") else None
        wtrace = ops.WGRAD_TRACE = [] if cfg["release"] else None
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = self.step()
        self.fence()
        dt = time.perf_counter() - t0
        ops.CONV_TRACE = ops.ATTN_TRACE = ops.WGRAD_TRACE = None
        if self.world > 1:
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        enc_ms = sum(s.elapsed_time(e) for s, e in self.ev) / max(len(self.ev), 1)
        frames = cfg["batch"] * cfg["length"]
        flops = ir50_forward_flops(cfg["hw"]) * frames
        achieved = flops / (enc_ms * 1e-3) / 1e12
        peak, peak_note = peak_tflops(cfg["precision"])
        # per kernel variant: algorithmic FLOPs of its launches / their HIP-event durations (launch stream), over the timed steps
        kernels = {}
        f16 = "true" if cfg["precision"] == "fp16" else "false"
        for tile, fl, e0, e1, nbytes in (trace or []):
            k = kernels.setdefault(KERNEL_NAMES.get(tile, f"conv tile {tile}").replace("{f16}", f16),
                                   {"launches": 0, "flops": 0.0, "ms": 0.0, "algo_bytes": 0.0})
            k["launches"] += 1
            k["flops"] += fl
            k["algo_bytes"] += nbytes
            k["ms"] += e0.elapsed_time(e1)
        for k in kernels.values():
            k["launches_per_step"] = k["launches"] / steps
            k["avg_launch_ms"] = k["ms"] / k["launches"]
            k["flops_per_launch"] = k["flops"] / k["launches"]
            k["achieved_tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12
            k["frac"] = k["achieved_tflops"] / peak
            k["algorithmic_bytes_per_launch"] = k["algo_bytes"] / k["launches"]
            k["algorithmic_gbps"] = k["algo_bytes"] / (k["ms"] * 1e-3) / 1e9
            k["ms_per_step"] = k.pop("ms") / steps
            del k["flops"], k["algo_bytes"]
        dominant = max(kernels, key=lambda n: kernels[n]["ms_per_step"]) if kernels else None
        what = {"fp32": "whole IR-50 forward on cer::conv_igemm_kernel (v_mfma_f32_32x32x2_f32)",
                "bf16x3": "whole IR-50 forward (51 bf16x3 implicit-GEMM convs + head FC, the input layer as a two-pass direct convolution, batch-statistics BatchNorm passes)",
                }.get(cfg["precision"], "whole IR-50 forward (51 narrow implicit-GEMM convs + head FC, the input layer as a two-pass direct convolution, batch-statistics "
                                        "BatchNorm passes)")
        span = {"what": what + ": algorithmic IR-50 FLOPs of the step / HIP-event span of the encoder forward",
                "achieved": achieved, "frac": achieved / peak, "ms_per_step_in_kernel": enc_ms, "algorithmic_flops_per_step": flops}
        if dominant is not None:
            kd = kernels[dominant]
            roofline = {"bound": "mfma", "kernel": dominant, "achieved": kd["achieved_tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": kd["frac"], "peak_note": peak_note,
                        "avg_launch_ms": kd["avg_launch_ms"], "launches_per_step": kd["launches_per_step"],
                        "algorithmic_flops_per_launch": kd["flops_per_launch"],
                        "definition": "algorithmic FLOPs (2*M*Cout*Cin*KH*KW) of this kernel's launches / their summed HIP-event "
                                      "durations on the launch stream over the timed steps",
                        "traffic": measured_traffic(cfg, dominant),
                        "traffic_note": "HBM bytes per launch of this kernel (rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes, "
                                        "FETCH x2 per the gfx950 correction, profiles/round3_traffic_*.json, keyed by configuration "
                                        "and kernel-source hash); algorithmic bytes per launch = input + output tensors + weights "
                                        "(+ residual)",
                        "algorithmic_bytes_per_launch": kd["algorithmic_bytes_per_launch"],
                        "traffic_per_step_all_convs": measured_traffic(cfg),
                        "kernel_source_sha": kernel_source_sha(),
                        "all_kernels": kernels, "encoder_span": span}
        else:
            roofline = {"bound": "mfma", "kernel": "cer::conv_igemm_kernel (IR-50 forward: 52 implicit-GEMM convs + head FC, "
                                                   "v_mfma_f32_32x32x2_f32)", "achieved": achieved, "peak": peak,
                        "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note, "traffic": None,
                        "algorithmic_flops_per_step": flops, "ms_per_step_in_kernel": enc_ms}
        if wtrace:
            # conv2d_wgrad_b3_kernel by layer family: algorithmic FLOPs 2*R*Cout*Cin*taps / HIP-event durations of the launches
            fam = {}
            for cout, cin, taps, fl, e0, e1 in wtrace:
                f = fam.setdefault(f"{cin}->{cout} {'3x3' if taps == 9 else '1x1'}", {"launches": 0, "flops": 0.0, "ms": 0.0})
                f["launches"] += 1
                f["flops"] += fl
                f["ms"] += e0.elapsed_time(e1)
            tot_f = sum(f["flops"] for f in fam.values())
            tot_ms = sum(f["ms"] for f in fam.values())
            wide = [f for n_, f in fam.items() if int(n_.split("->")[0]) >= 256 and "3x3" in n_]
            for f in fam.values():
                f["achieved_tflops"] = f["flops"] / (f["ms"] * 1e-3) / 1e12
                f["frac"] = f["achieved_tflops"] / (BF16_MFMA_PEAK_TFLOPS / 3.0)
                f["ms_per_step"] = f.pop("ms") / steps
                f["launches_per_step"] = f.pop("launches") / steps
                del f["flops"]
            roofline["wgrad"] = {"kernel": "cer::conv2d_wgrad_b3_kernel (bf16x3 operands, transposed LDS reads)", "families": fam,
                                 "achieved": tot_f / (tot_ms * 1e-3) / 1e12, "frac": tot_f / (tot_ms * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 3.0),
                                 "ms_per_step": tot_ms / steps, "peak": BF16_MFMA_PEAK_TFLOPS / 3.0, "unit": "TFLOP/s"}
            roofline["wgrad_frac"] = roofline["wgrad"]["frac"]
            roofline["wgrad_ms_per_step"] = tot_ms / steps
            if wide:
                roofline["wgrad_frac_256_512ch"] = min(f["frac"] for f in wide)
        if torch.cuda.is_available():
            roofline["peak_hbm_allocated_gb"] = torch.cuda.max_memory_allocated() / 1e9
        if atrace:
            att = {}
            for kind, fl, e0, e1 in atrace:
                a = att.setdefault(kind, {"launches": 0, "flops": 0.0, "ms": 0.0})
                a["launches"] += 1
                a["flops"] += fl
                a["ms"] += e0.elapsed_time(e1)
            for kind, a in att.items():
                a["kernel"] = f"cer::attention_{kind.split('_')[0]}_kernel<128> (exact fp32 MFMA, flash style" + \
                    (f"; the {kind.split('_')[1]}-token launches: cfg3's final stage, split-stream variant)" if "_" in kind else
                     "; the per-clip launches: 32 tokens x 32 clips, latency-bound by size)")
                a["achieved"] = a["flops"] / (a["ms"] * 1e-3) / 1e12
                a["peak"], a["unit"], a["bound"] = FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", "mfma"
                a["frac"] = a["achieved"] / FP32_MFMA_PEAK_TFLOPS
                a["launches_per_step"] = a["launches"] / steps
                a["ms_per_step"] = a.pop("ms") / steps
                a["algorithmic_flops_per_step"] = a.pop("flops") / steps
            att["definition"] = ("algorithmic FLOPs: forward 4*Sq*Sk*D, backward 8*Sq*Sk*D (dV, dP, dQ, dK; the recomputed "
                                 "S = QK^T is not counted) per (batch, head), / HIP-event durations of the launches")
            roofline["attention"] = att
        mods = cfg["modalities"]
        enc_txt = ("VGGish (log-mel of the clip's PCM, one 0.96 s example per frame)" +
                   (" and BERT-base (64-token sentence)" if "bert" in mods else "") + " run on the GPU inside the step") \
            if cfg["encoders"] == "on" else "vggish/bert as pre-computed per-frame features (as the reference trainer feeds them)"
        return {
            "value": self.world * cfg["batch"] * steps / dt, "unit": "clips/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "dtype": dtype_string(cfg["precision"]),
            "config": {"workload": f"{cfg['model']} {'+'.join(mods)} training step: frozen IR-50 forward on {cfg['batch']}x{cfg['length']} "
                                   f"frames of {cfg['hw']}x{cfg['hw']} + TCN/fusion/classifier forward+backward + CE + fused Nesterov "
                                   f"SGD; " + enc_txt,
                       "model": cfg["model"], "modalities": mods, "encoders_on_gpu": cfg["encoders"] == "on",
                       "clips_per_gpu": cfg["batch"], "global_batch": cfg["batch"] * self.world, "frames_per_clip": cfg["length"],
                       "frame_hw": cfg["hw"], "n_classes": cfg["n_cls"], "released_encoder_groups": cfg["release"],
                       "released_units_activation_memory": (getattr(self.model.spatial["visual"].backbone, "_act_mem", cfg["act_mem"])
                                                            if cfg["release"] else None),
                       "trainable_parameters": int(sum(p.numel() for p in self.ddp.params)), "conv_precision": cfg["precision"],
                       "parallelism": f"dp{self.world} over clips, flat-bucket RCCL all-reduce", "loss": float(loss.item())},
            "roofline": roofline}

    def close(self):
        for h in self._hooks:
            h.remove()


def compact(res):
    """A secondary measurement as it is nested under roofline.other_configs: the headline figures without the per-kernel list."""
    r = res["roofline"]
    out = {"value": res["value"], "unit": res["unit"], "ms_per_step": res["ms_per_step"], "steps": res["steps"],
           "warmup": res["warmup"], "dtype": res["dtype"], "config": res["config"],
           "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "peak_note", "avg_launch_ms",
                                          "launches_per_step", "algorithmic_bytes_per_launch", "traffic",
                                          "traffic_per_step_all_convs") if k in r}}
    if "encoder_span" in r:
        out["roofline"]["encoder_span"] = {k: r["encoder_span"][k] for k in ("achieved", "frac", "ms_per_step_in_kernel")}
    if "all_kernels" in r:
        out["roofline"]["all_kernels"] = {n: {"frac": k["frac"], "achieved_tflops": k["achieved_tflops"], "ms_per_step": k["ms_per_step"],
                                              "launches_per_step": k["launches_per_step"]} for n, k in r["all_kernels"].items()}
    if "attention" in r:
        out["roofline"]["attention"] = r["attention"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--hw", type=int, default=224, help="frame size: 224 (north-star shape) or 40 (reference crop)")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--length", type=int, default=32)
    ap.add_argument("--n-cls", type=int, default=7, help="7 (MELD, C-EXPR-DB) or 8 (C-EXPR-DB with use_other_class)")
    ap.add_argument("--model", choices=["LFAN", "JMT", "MT", "CAN"], default="LFAN")
    ap.add_argument("--modalities", default=None, help="comma list; default video,vggish,bert (LFAN) / video,vggish (JMT, MT, CAN)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra configurations (other precisions, cfg5, cfg3; N = 1 only)")
    ap.add_argument("--precision", choices=["bf16x3", "fp32", "bf16", "fp16"], default="bf16x3",
                    help="IR-50 conv kernels: bf16x3 = split hi/lo bf16 operands, 3 bf16 MFMAs per product (logit error ~1e-6); "
                         "fp32 = exact fp32 MFMA; bf16 / fp16 = narrow storage, one MFMA per product, fp32 accumulate (what the "
                         "reference's --amp recipe computes; BASELINE cfg5)")
    ap.add_argument("--release", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="gradual-release groups of the reference's ResnetParamControl to un-freeze before timing "
                         "(0 = as the reference trains: encoder frozen; 1 = output layer; 2 = + stage 4; 3 = + half of stage 3; "
                         "4 = extension: the whole IR-50 trains, forward + backward through every unit and the stem)")
    ap.add_argument("--act-mem", choices=["auto", "raw", "recompute", "recompute16"], default="auto",
                    help="released encoder units: raw = keep the raw conv results for the backward (410 MB per 224x224 frame); "
                         "recompute = keep unit inputs as one fp16 plane (62 MB per frame) and re-run the unit's convs in the "
                         "backward (bit-identical gradients); auto = raw when it fits the free device memory, else recompute")
    ap.add_argument("--encoders", choices=["on", "off"], default="on",
                    help="on: VGGish (log-mel from PCM) and BERT (64 tokens) run on the GPU inside the step; "
                         "off: pre-computed per-frame features, as the reference trainer feeds them")
    a = ap.parse_args()

    from feature_vs_text_compound_emotion_amd.data_parallel import init_process_group_from_env

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

    mods = a.modalities.split(",") if a.modalities else (ALL_MODS if a.model == "LFAN" else ["video", "vggish"])
    cfg = {"model": a.model, "modalities": mods, "hw": a.hw, "batch": a.batch, "length": a.length, "n_cls": a.n_cls,
           "precision": a.precision, "release": a.release, "encoders": a.encoders}
    cfg["act_mem"] = a.act_mem      # "auto": the encoder picks raw / recompute from the free device memory (IR50.activation_memory)
    wl = Workload(cfg, rank, world, dev)
    res = wl.measure(a.steps, a.warmup)
    wl.close()
    del wl
    torch.cuda.empty_cache()

    others = {}
    if world == 1 and not a.no_alt and a.model == "LFAN" and a.release == 0:
        def leg(name, steps=5, warmup=1, **over):
            c = dict(cfg)
            c.update(over)
            w = Workload(c, rank, world, dev)
            others[name] = compact(w.measure(steps, warmup))
            w.close()
            del w
            torch.cuda.empty_cache()
        for prec in ("fp32", "fp16", "bf16"):
            if prec != a.precision:
                leg(f"same workload, conv_precision {prec}", precision=prec)
        # BASELINE cfg5: tri-modal C-EXPR-DB config, 64-frame clips, 8 classes (7 + Other), bf16 storage
        leg("cfg5: tri-modal, 64-frame clips, 8 classes, bf16 storage", steps=5, warmup=1, precision="bf16", length=64, n_cls=8)
        # BASELINE cfg3: vision + VGGish, transformer (JMT) head whose final attention runs over L*B = 1024 tokens x 6 stacks
        leg("cfg3: video+vggish, JMT head (MFMA attention over 1024 tokens)", steps=5, warmup=1, model="JMT",
            modalities=["video", "vggish"])

    # the side configurations as FLAT scalars of the driver-visible record (the driver's parser keeps scalars of `roofline`
    # and drops nested objects): cfg5 / fp16 / bf16 / fp32 / cfg3 and the whole-encoder figures
    flat = {}
    r0 = res["roofline"]
    if "encoder_span" in r0:
        flat["encoder_span_frac"] = r0["encoder_span"]["frac"]
        flat["encoder_span_ms"] = r0["encoder_span"]["ms_per_step_in_kernel"]
    for name, o in others.items():
        tag = name.split(":")[0] if name.startswith("cfg") else name.rsplit(" ", 1)[1]
        flat[f"{tag}_clips_s"] = o["value"]
        flat[f"{tag}_ms_per_step"] = o["ms_per_step"]
        ro = o["roofline"]
        flat[f"{tag}_dominant_frac"] = ro.get("frac")
        if "encoder_span" in ro:
            flat[f"{tag}_encoder_frac"] = ro["encoder_span"]["frac"]
        for kn, kv in ro.get("all_kernels", {}).items():
            if kn.startswith("cer::conv_n16_patch_kernel<64") or kn.startswith("cer::conv_n16_p64_kernel"):
                flat[f"{tag}_patch64_frac"] = kv["frac"]
        for kind, av in ro.get("attention", {}).items():
            if isinstance(av, dict) and "frac" in av:
                flat[f"{tag}_attention_{kind}_frac"] = av["frac"]
    if rank == 0:
        res["roofline"].update(flat)
        out = {"metric": "training clips/sec (32-frame tri-modal clip)", "value": res["value"], "unit": "clips/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": res["dtype"], "data": "synthetic",
               "config": res["config"], "roofline": res["roofline"], "extra": flat}
        if others:
            out["roofline"]["other_configs"] = others
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.hw, a.length, a.encoders == "on")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
