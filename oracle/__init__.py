"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference's
arithmetic for the multimodal feature-extraction + temporal-fusion path.  It is
written in functional style over a flat ``state_dict`` that uses the
reference's parameter names, so a reference checkpoint (or a seeded random
one) drives both the oracle and the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package
(``feature_vs_text_compound_emotion_amd``) never does: it fails loudly when the
HIP library is missing instead of falling back to this code.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4), so every function here is pinned against the reference's
own ``models/`` package imported in the build container by
``tools/gen_golden.py``; the resulting small fixtures are committed under
``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks the oracle
against them without needing the reference.
"""
from .ir50 import ir50_forward, ir50_block_plan, IR50_STAGES  # noqa: F401
from .tcn import tcn_forward, weight_norm_weight  # noqa: F401
from .fusion import lfan_fusion_forward  # noqa: F401
from .lfan import lfan_forward, cross_entropy_mean, sgd_nesterov_step  # noqa: F401
from .vggish import vggish_forward, waveform_to_examples, wav_int16_to_examples, log_mel_spectrogram  # noqa: F401,E402
from .bert import bert_hidden_states, bert_token_features, exclude_padding  # noqa: F401,E402
from .jmt import jmt_forward, can_forward, jmt_fusion  # noqa: F401,E402
