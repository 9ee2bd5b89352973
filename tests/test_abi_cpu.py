"""CPU-side checks of the drop-in boundary: the C-ABI library loads here (no GPU needed for
dlopen) and exports exactly what include/cer_hip.h declares; host modules mirror the
reference's state-dict layout."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "cer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cer_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from feature_vs_text_compound_emotion_amd import _lib
    from feature_vs_text_compound_emotion_amd.build import build
    build(verbose=False)
    lib = _lib.load()
    declared = _header_functions()
    assert declared == _lib.exported_symbols()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert lib.cer_version() >= 100
    assert lib.cer_conv_kpad(3, 3, 3) == 32 and lib.cer_conv_kpad(3, 3, 64) == 576


def test_invalid_arguments_report_errors_without_a_gpu():
    from feature_vs_text_compound_emotion_amd import _lib
    lib = _lib.load()
    rc = lib.cer_l2norm_rows(None, None, 0, 0, None)
    assert rc == -1 and b"l2norm_rows" in lib.cer_last_error()
    d = _lib.ConvDesc()
    assert lib.cer_conv2d_fwd(ctypes.byref(d), *([None] * 12), 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from feature_vs_text_compound_emotion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    from feature_vs_text_compound_emotion_amd import ops
    with pytest.raises(ValueError):
        ops.l2norm_rows(torch.zeros(2, 4))


def test_lfan_state_dict_layout_matches_reference_spec():
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import LFAN
    mods = ["video", "vggish", "bert"]
    m = LFAN(backbone_settings={}, output_dim=7, task="CLASSIFICATION", modality=mods, example_length=8,
             tcn_channel=synth.TCN_CHANNELS, root_dir="", device="cpu")
    m.init(load_backbone=False)
    spec, alias = synth.lfan_spec(mods)
    sd = m.state_dict()
    assert set(sd) == set(spec) | set(alias)
    for k, (shape, _) in spec.items():
        assert tuple(sd[k].shape) == shape, k
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert not any(n.startswith("spatial.") for n in trainable)
    assert sum(p.numel() for n, p in m.named_parameters() if p.requires_grad) == 5002503
    assert "visual" in m.spatial  # base/parameter_control.py:85-96 contract
    import copy
    copy.deepcopy(m)  # trainer.py:656,705 deep-copies the model


def test_lfan_forward_without_gpu_raises():
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import LFAN
    m = LFAN(backbone_settings={}, output_dim=7, task="CLASSIFICATION", modality=["vggish"], example_length=4,
             tcn_channel=synth.TCN_CHANNELS, root_dir="", device="cpu")
    m.init()
    with pytest.raises((ValueError, RuntimeError)):
        m({"vggish": torch.zeros(1, 1, 4, 128)})
