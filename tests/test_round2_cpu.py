"""CPU-side checks of round-2 host logic: LDS swizzle tables (brute force over the bank model), scores derived from
confusion counts vs the numpy mirror of the reference's metrics, bench.py helpers."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lds_swizzles_are_conflict_free_under_the_ds_read_b128_bank_model():
    cs = _load(os.path.join(ROOT, "tools", "check_swizzle.py"), "check_swizzle")
    for base in range(0, 256, 16):
        for kk in (0, 1):
            assert cs.worst_way(lambda l: cs.n16_fragment_addr(l, base, kk)) == 1
        assert cs.worst_way(lambda l: cs.b3_fragment_addr(l, base)) == 1
    assert cs.n16_dma_image_is_a_permutation()
    for r in range(16):
        for kh in range(3):
            for kw in range(3):
                assert cs.worst_way(lambda l: cs.b3_patch_fragment_addr(l, r, kh, kw)) == 1
                for kk in (0, 1):
                    assert cs.worst_way(lambda l: cs.patch_fragment_addr(l, r, kh, kw, kk)) == 1
    # the constants compiled into csrc/conv_n16_patch.hip and csrc/conv_b3_patch.hip
    assert cs.patch_table_constant() == 0xd92dad912240
    assert sum(f << (2 * i) for i, f in enumerate(cs.B3_PATCH_F)) == 0xaaa00a00
    # a linear (unswizzled) image of 128-byte rows is a 4-way conflict: the model does discriminate
    assert cs.worst_way(lambda l: (l & 15) * 128 + ((l >> 4) << 4)) == 4
    # 1-D window kernels: fragment rows start at ANY row (tap shift kh * W + kw for arbitrary W): one swizzle per row width
    for first_row in range(64):
        assert cs.worst_way(lambda l: cs.b3_win_fragment_addr(l, first_row)) == 1
        for kk in (0, 1):
            assert cs.worst_way(lambda l: cs.n16_win_fragment_addr(l, first_row, kk)) == 1


def test_release_plan_accepts_suffixes_of_the_body_and_fails_loudly_otherwise():
    """IR50._release_plan only reads requires_grad flags: the reference's three release groups, the whole-encoder extension
    (BASELINE configs[1]) and the rejected combinations, without a GPU."""
    import pytest
    import torch
    from feature_vs_text_compound_emotion_amd.parameter_control import ResnetParamControl
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    vb = VisualBackbone(use_pretrained=False, head_hw=5)
    bb = vb.backbone

    def freeze():
        for p in vb.parameters():
            p.requires_grad = False

    freeze()
    assert bb._release_plan() is None
    pc = ResnetParamControl(trainer=None, release_count=3)
    expect = [24, 21, 18]          # head only; + stage 4 (units 21-23); + the last three units of stage 3 (parameters 142..162)
    for first in expect:
        pc.release_param({"visual": vb})
        assert bb._release_plan() == first
        assert not bb._stem_released()
    for p in bb.body.parameters():
        p.requires_grad = True
    assert bb._release_plan() == 0 and not bb._stem_released()
    for p in bb.input_layer.parameters():
        p.requires_grad = True
    assert bb._release_plan() == 0 and bb._stem_released()
    with torch.no_grad():
        assert bb._release_plan() is None and not bb._stem_released()
    freeze()
    for p in list(bb.output_layer.parameters()) + list(bb.body[5].parameters()):
        p.requires_grad = True
    with pytest.raises(NotImplementedError, match="suffix of the body"):
        bb._release_plan()
    freeze()
    for p in list(bb.output_layer.parameters()) + list(bb.input_layer.parameters()):
        p.requires_grad = True
    with pytest.raises(NotImplementedError, match="together with the whole body"):
        bb._release_plan()
    freeze()
    bb.body[23].res_layer[1].weight.requires_grad = True
    with pytest.raises(NotImplementedError, match="whole units"):
        bb._release_plan()
    freeze()
    for p in bb.body[23].parameters():
        p.requires_grad = True
    with pytest.raises(NotImplementedError, match="whole output layer"):
        bb._release_plan()


def test_scores_from_confusion_counts_equal_the_label_list_definitions():
    from feature_vs_text_compound_emotion_amd import metrics
    from feature_vs_text_compound_emotion_amd.eval_device import scores_from_confusion
    rng = np.random.default_rng(3)
    for n_cls, n in ((7, 500), (8, 40), (7, 3)):
        trg, prd = rng.integers(0, n_cls - 1, n).tolist(), rng.integers(0, n_cls, n).tolist()  # a class that never is a target
        cm = np.zeros((n_cls, n_cls), dtype=np.int64)
        for t, p in zip(trg, prd):
            cm[t, p] += 1
        s = scores_from_confusion(cm)
        f1s, macro = metrics.compute_f1_score(trg, prd, metrics.MACRO_F1)
        assert np.allclose(s["f1_per_class"], f1s) and abs(s["macro_f1"] - macro) < 1e-12
        assert abs(s["weighted_f1"] - metrics.compute_f1_score(trg, prd, metrics.W_F1)[1]) < 1e-12
        assert abs(s["accuracy"] - metrics.compute_class_acc(trg, prd)) < 1e-9
        assert np.allclose(s["confusion"], metrics.compute_confusion_matrix(trg, prd))


def test_bench_helpers():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_module")
    assert bench.peak_tflops("bf16")[0] == 2500.0 and bench.peak_tflops("fp16")[0] == 2500.0
    assert abs(bench.peak_tflops("bf16x3")[0] - 2500.0 / 3) < 1e-9 and bench.peak_tflops("fp32")[0] == 157.3
    assert abs(bench.ir50_forward_flops(40) - 6.0545e9) / 6.0545e9 < 1e-3
    assert abs(bench.ir50_forward_flops(224) - 189.869e9) / 189.869e9 < 1e-3
    sha = bench.kernel_source_sha()
    assert len(sha) == 16 and sha == bench.kernel_source_sha()
    # a committed traffic profile is only reported for the kernel sources it was taken on
    cfg = {"precision": "bf16x3", "hw": 224, "length": 32, "batch": 32, "encoders": "on"}
    import json
    expect = None
    for rnd in ("round3", "round2"):          # the newest profile taken on these sources wins
        t = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_traffic_bf16x3_hw224_L32.json")))
        if t["kernel_source_sha"] == sha:
            expect = t["hbm_bytes_per_step"]
            break
    assert bench.measured_traffic(cfg) == expect
    assert bench.measured_traffic(dict(cfg, release=4)) is None     # a profile of the frozen-encoder step says nothing about that one


def test_space_to_depth_weight_order_and_layout_describe_the_stride2_conv():
    """Host side of the space-to-depth stride-2 kernels: cer_conv_s2d_k_order is a permutation of the 9*C weight columns in
    (phase, chunk, shift) step order, and a stride-1 2x2 conv over the space-to-depth tensor with those columns IS the
    3x3 / stride 2 / pad 1 conv (torch on the CPU; the kernels are tested on the GPU)."""
    import ctypes

    import torch
    import torch.nn.functional as F

    from feature_vs_text_compound_emotion_amd import _lib, ops
    lib = _lib.load()
    for cin, chunk in ((64, 32), (128, 64), (128, 32)):
        order = (ctypes.c_int32 * (9 * cin))()
        assert lib.cer_conv_s2d_k_order(cin, chunk, order) == 0
        order = np.array(list(order))
        assert sorted(order.tolist()) == list(range(9 * cin))
        taps = order.reshape(-1, chunk)[:, 0] // cin                      # the filter tap of every step
        per_phase = [4 * cin // chunk, 2 * cin // chunk, 2 * cin // chunk, cin // chunk]
        bounds = np.cumsum([0] + per_phase)
        for ph, allowed in enumerate(([0, 2, 6, 8], [1, 7], [3, 5], [4])):
            assert set(taps[bounds[ph]:bounds[ph + 1]].tolist()) == set(allowed)
    assert lib.cer_conv_s2d_k_order(96, 32, (ctypes.c_int32 * (9 * 96))()) != 0    # an odd number of chunks per phase
    # the algebra: phase images + per-phase shifts reproduce the strided conv
    g = torch.Generator().manual_seed(3)
    n, c, h, w, co = 2, 4, 8, 6, 5
    x, wt = torch.randn(n, h, w, c, generator=g), torch.randn(co, c, 3, 3, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt, None, 2, 1).permute(0, 2, 3, 1)
    xs = ops.space_to_depth(x)                                            # [n, h/2, w/2, 4c], blocks P11 | P10 | P01 | P00
    assert tuple(xs.shape) == (n, h // 2, w // 2, 4 * c)
    shifts = {0: [(0, 0, 0), (0, 1, 2), (1, 0, 6), (1, 1, 8)], 1: [(0, 1, 1), (1, 1, 7)], 2: [(1, 0, 3), (1, 1, 5)], 3: [(1, 1, 4)]}
    out = torch.zeros_like(ref)
    pad = F.pad(xs, (0, 0, 1, 0, 1, 0))                                   # one zero row above / column to the left
    for blk, lst in shifts.items():
        img = pad[..., blk * c:(blk + 1) * c]
        for kh, kw, tap in lst:                                           # shift (kh - 1, kw - 1) of the phase image
            sl = img[:, kh:kh + h // 2, kw:kw + w // 2]
            out += torch.einsum("nhwc,oc->nhwo", sl, wt[:, :, tap // 3, tap % 3])
    assert (out - ref).abs().max().item() < 1e-5


def test_reciprocal_division_of_the_kernel_prologues_is_exact():
    """csrc/conv_common.h FastDiv / fdiv (block -> patch / tile, pixel -> image row / column in the window and patch kernels):
    magic = 2^32 / d + 1, q = umulhi(x, magic), q -= (q * d > x).  The arithmetic restated on 64-bit integers: exact for every
    x < 2^31 (M = N H W is an int) and every divisor the launchers build, and q * d never wraps 32 bits."""
    rng = np.random.default_rng(0)
    divisors = list(range(1, 300)) + [392, 784, 3136, 12544, 50176, 65535, 65536, 100003, 2 ** 20 + 7, 2 ** 24 - 3, 2 ** 30 + 1]
    for d in divisors:
        magic = np.uint64(((1 << 32) // d + 1) & 0xffffffff)
        mult = np.arange(1, 3000, dtype=np.uint64) * np.uint64(d)
        xs = np.concatenate([rng.integers(0, 2 ** 31, 20000, dtype=np.uint64), np.arange(0, 5000, dtype=np.uint64),
                             mult % np.uint64(2 ** 31), (mult - np.uint64(1)) % np.uint64(2 ** 31),
                             np.array([2 ** 31 - 1, 2 ** 31 - 2], dtype=np.uint64)])
        if d == 1:
            q = xs.copy()
        else:
            q = (xs * magic) >> np.uint64(32)
            assert (q * np.uint64(d) < np.uint64(2 ** 32)).all()          # the 32-bit product of the correction step does not wrap
            q = q - (q * np.uint64(d) > xs)
        assert (q == xs // np.uint64(d)).all(), d


def test_direct_epilogue_weight_row_permutation_makes_tile_pairs_contiguous():
    """csrc/conv_common.h epi_cout_of_row: MFMA tile a, tile row 4 kg + r of a wave's 16 TC weight rows holds cout
    32 (a / 2) + 8 kg + 4 (a % 2) + r.  A bijection on the wave's span, and the eight accumulators a lane holds of a tile pair
    (two tiles x four rows) are eight CONSECUTIVE couts starting at 32 j + 8 kg -- what lets the epilogue store 16 bytes per lane."""
    def cout_of_row(rho):
        return ((rho >> 5) << 5) + (((rho >> 2) & 3) << 3) + (((rho >> 4) & 1) << 2) + (rho & 3)
    for tc in (2, 4, 8):
        span = 16 * tc
        assert sorted(cout_of_row(r) for r in range(span)) == list(range(span))
        for j in range(tc // 2):
            for kg in range(4):
                got = [cout_of_row(16 * (2 * j + half) + 4 * kg + r) for half in (0, 1) for r in range(4)]
                assert got == list(range(32 * j + 8 * kg, 32 * j + 8 * kg + 8))
