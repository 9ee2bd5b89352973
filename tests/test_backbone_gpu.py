"""IR-50 vision encoder on the HIP kernels vs the oracle and the reference-recorded fixtures."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import golden  # noqa: E402

EMB_TOL = 1e-4  # unit-norm embeddings; fp32 MFMA vs fp32 CPU differ only by summation order


def _build(sd_prefixless, head_hw):
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    vb = VisualBackbone(use_pretrained=False, head_hw=head_hw)
    vb.load_state_dict(sd_prefixless, strict=True)
    return vb.cuda().eval()


def test_state_dict_keys_match_reference_layout():
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    vb = VisualBackbone(use_pretrained=False)
    spec = synth.visual_backbone_spec("", 5)
    assert len(vb.state_dict()) == 351
    assert set(vb.state_dict()) == set(spec)
    for k, v in vb.state_dict().items():
        assert tuple(v.shape) == spec[k][0], k


def test_embedding_matches_reference_fixture():
    from feature_vs_text_compound_emotion_amd import synth
    g = golden("visual_backbone_eval.npz")
    n, hw, wseed, dseed = [int(v) for v in g["meta"]]
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=wseed)
    frames = torch.randn(n, 3, hw, hw, generator=torch.Generator().manual_seed(dseed))
    vb = _build(vsd, hw // 8)
    with torch.no_grad():
        emb = vb(frames.cuda()).cpu().numpy()
    assert np.abs(emb - g["emb"]).max() < EMB_TOL


@pytest.mark.parametrize("n,hw", [(1, 40), (37, 40), (2, 224), (3, 64)])
def test_embedding_matches_oracle(n, hw):
    import oracle
    from feature_vs_text_compound_emotion_amd import synth
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=11)
    frames = torch.randn(n, 3, hw, hw, generator=torch.Generator().manual_seed(n + hw))
    with torch.no_grad():
        ref = oracle.ir50_forward(frames, vsd, "backbone.")
        emb = _build(vsd, hw // 8)(frames.cuda()).cpu()
    assert (emb - ref).abs().max().item() < EMB_TOL
    assert (emb.norm(dim=1) - 1).abs().max().item() < 1e-5


def test_wrong_frame_size_raises():
    from feature_vs_text_compound_emotion_amd import synth
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", 5), seed=1)
    vb = _build(vsd, 5)
    with pytest.raises(RuntimeError):
        vb(torch.zeros(1, 3, 48, 48, device="cuda"))
