"""Where the narrow-mode parity bars come from, shown on the CPU oracle alone (no kernel involved).

Round 2 loosened two GPU assertions after red runs and explained them with "almost equal embeddings / a tiny BatchNorm1d
spread"; the round-2 verdict asked for that explanation to be proven or retracted.  It is retracted -- the embeddings of the
synthetic frames are not almost equal (pairwise cosine down to -0.2 in train mode) -- and replaced by the two mechanisms
these tests demonstrate:

1. FORWARD: in eval mode the synthetic running statistics of ``bn.video`` (variance ~ U(0.5, 1.5)) are ~1000x the actual
   variance of the video temporal net's output (std 0.03), so the video modality enters the logits scaled by ~0.03 and its
   storage error is invisible; ``model.train()`` normalises by the BATCH statistics and the embedding error enters at full
   weight.  16-bit storage alone (oracle/narrow.py: every tensor rounded once, everything else fp32) then gives
   ~2e-2 (bf16) / ~3e-3 (fp16) on the cfg5 batch, and the reference's own narrow arithmetic (torch autocast,
   trainer.py:367) gives 2-3x that.  SURVEY section 7's 2e-2 for bf16 came from an eval-mode measurement; in the mode the
   reference trains in it is the format's floor, not a margin.
2. UPDATE: a LeakyReLU pre-activation of the video temporal net that changes sign under the perturbation multiplies its
   gradient by 100 behind a batch-statistics BatchNorm1d; on one batch seed that moves the two-step update by ~40 %, on
   another by < 1 % with the same weights; with biases at the scale of the signal (``synth.lfan_state_dict(conditioned=
   True)``) the response is a few percent on both.
"""
import torch

from helpers import MODS, trainable_names
from feature_vs_text_compound_emotion_amd import synth
from oracle import ir50_forward, lfan_forward
from oracle.lfan import cross_entropy_mean, sgd_nesterov_step
from oracle.narrow import autocast_lfan_forward, ir50_forward_narrow_storage
import oracle.lfan as oracle_lfan

VB = "spatial.visual.backbone."


def _with_embedding(e, fn):
    """run ``fn`` with the oracle's IR-50 replaced by a precomputed embedding"""
    orig = oracle_lfan.ir50_forward
    oracle_lfan.ir50_forward = lambda *a, **k: e
    try:
        return fn()
    finally:
        oracle_lfan.ir50_forward = orig


def test_storage_floor_of_the_train_mode_logits_and_the_autocast_yardstick():
    b, length, hw, n_cls = 2, 64, 40, 8          # the batch tests/test_narrow_gpu.py::test_cfg5_* runs on the GPU
    sd = synth.lfan_state_dict(MODS, n_cls=n_cls, head_hw=hw // 8, seed=0)
    x, _ = synth.make_clip_batch(MODS, b, length, hw=hw, seed=4321, n_cls=n_cls)
    frames = x["video"].reshape(-1, 3, hw, hw)
    with torch.no_grad():
        ref = lfan_forward(x, sd, MODS, train=True, backbone_train=True)
        ref_eval = lfan_forward(x, sd, MODS, train=False)
        e32 = ir50_forward(frames, sd, VB, train=True)
        # the embeddings are NOT almost equal (the retracted explanation)
        cos = torch.nn.functional.cosine_similarity(e32[:, None], e32[None], dim=-1)
        assert cos.min().item() < 0.5
        for dt, lo, hi in ((torch.bfloat16, 1.2e-2, 3.5e-2), (torch.float16, 1.5e-3, 4.5e-3)):
            e = ir50_forward_narrow_storage(frames, sd, VB, dt)
            rel = ((e - e32).norm() / e32.norm()).item()
            err = (_with_embedding(e, lambda: lfan_forward(x, sd, MODS, train=True, backbone_train=True)) - ref).abs().max().item()
            ac = (autocast_lfan_forward(x, sd, MODS, dt, train=True, backbone_train=True) - ref).abs().max().item()
            ac_eval = (autocast_lfan_forward(x, sd, MODS, dt, train=False) - ref_eval).abs().max().item()
            print(f"\n[{dt}] storage-only emulation: embedding rel L2 {rel:.2e}, train-mode |logit err| {err:.2e}; "
                  f"reference autocast arithmetic: train {ac:.2e}, eval {ac_eval:.2e}")
            assert lo < err < hi                 # the format's floor on this batch
            assert ac > 1.5 * err                # the reference's own narrow recipe is further from fp32 than storage alone
            assert ac > 3 * ac_eval              # and train mode exposes what eval mode hides, for the reference too
            assert 0.5 * err < 1.3 * rel < 2.0 * err + 1e-3   # logit error ~ 1.3 x the relative embedding error


def _two_step_update(sd, alias, names, x, labels, hw, dtype):
    osd = {k: v.clone() for k, v in sd.items()}
    bufs = [None] * len(names)
    flips = 0
    for _ in range(2):
        params = [osd[n].clone().requires_grad_(True) for n in names]
        sds = dict(osd)
        sds.update(zip(names, params))
        for a, src in alias.items():
            sds[a] = sds[src]

        def fwd():
            return lfan_forward(x, sds, MODS, train=True, backbone_train=True)
        if dtype is None:
            logits = fwd()
        else:
            with torch.no_grad():
                e = ir50_forward_narrow_storage(x["video"].reshape(-1, 3, hw, hw), osd, VB, dtype)
            logits = _with_embedding(e, fwd)
        grads = torch.autograd.grad(cross_entropy_mean(logits, labels), params)
        newp, bufs = sgd_nesterov_step([p.detach() for p in params], list(grads), bufs)
        for n, p in zip(names, newp):
            osd[n] = p
        for a, src in alias.items():
            osd[a] = osd[src]
    return torch.cat([osd[n].reshape(-1) for n in names])


def test_update_cliff_is_a_leaky_relu_switch_and_conditioned_weights_remove_it():
    b, length, hw = 4, 8, 40                     # the batch of tests/test_dp_amp_gpu.py::test_reference_amp_wrapper_*
    alias = synth.lfan_spec(MODS)[1]
    out = {}
    for conditioned in (False, True):
        sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=hw // 8, seed=0, conditioned=conditioned)
        names = trainable_names(sd, alias)
        w0 = torch.cat([sd[n].reshape(-1) for n in names])
        for seed in (56, 57):
            x, labels = synth.make_clip_batch(MODS, b, length, hw=hw, seed=seed)
            w32 = _two_step_update(sd, alias, names, x, labels, hw, None)
            w16 = _two_step_update(sd, alias, names, x, labels, hw, torch.float16)
            out[conditioned, seed] = ((w16 - w32).norm() / (w32 - w0).norm()).item()
    print("\ntwo-step update, fp16 storage vs fp32, relative difference:", {k: f"{v:.2e}" for k, v in out.items()})
    assert out[False, 56] > 0.2                  # what the GPU measured (0.42) -- from a 2e-3 relative embedding error
    assert out[False, 57] < 0.05                 # same weights, another batch: no switch crossed
    assert out[True, 56] < 0.1 and out[True, 57] < 0.1
