"""On-GPU multimodal feature extraction: raw clip -> the per-frame features LFAN consumes.

In the reference, VGGish and BERT never run inside the training loop: an offline package writes
``vggish.npy [N,128]`` and ``bert.npy [N,768]`` per trial (SURVEY.md F5;
abaw5_pre_processing/base/audio.py:120-148, base/preprocessing.py:992-1018,
base/speech.py:185-251,690-738).  This module fuses that offline stage into the GPU hot path:

  audio : int16 PCM @16 kHz -> log-mel -> 0.96 s examples hopped by 1/fps -> VGGish -> one 128-d
          row per video frame (rows beyond the last example repeat it: compact_audio_feature's
          edge padding)
  text  : token ids (one padded sentence per clip) -> BERT sum-of-last-4 -> drop [CLS]/[SEP]/pad ->
          spread the tokens over the frames in contiguous blocks (align_word_embedding_new)
  video : frames pass through (IR-50 lives inside the model).
"""
import torch

from . import ops
from .audio_backbone import AudioBackbone
from .text_encoder import BertEncoderHIP


def align_tokens_to_frames(num_tokens, num_frames):
    """speech.py:690-738: drop tokens beyond the frame count, split the frame indices into
    ``num_tokens`` contiguous blocks (more_itertools.divide: the first ``len % n`` blocks get one
    extra frame) and give every frame of block j token j.  Returns a list of token indices per
    frame (empty when there is no token: the reference then emits zeros)."""
    n = min(num_tokens, num_frames)
    if n <= 0:
        return []
    q, r = divmod(num_frames, n)
    idx = []
    for j in range(n):
        idx += [j] * (q + 1 if j < r else q)
    return idx


class MultimodalFeatureExtractor(torch.nn.Module):
    def __init__(self, audio=None, text=None, fps=32):
        super().__init__()
        self.audio = audio if audio is not None else AudioBackbone()
        self.text = text if text is not None else BertEncoderHIP()
        self.fps = fps

    @torch.no_grad()
    def audio_features(self, pcm_int16, num_frames, sample_rate=16000):
        """pcm [B,S] int16 -> [B,1,L,128]."""
        ex = self.audio.backbone.wav_int16_to_examples(pcm_int16, sample_rate, 0.96, 1.0 / self.fps)
        b, n = ex.shape[:2]
        use = min(n, num_frames)
        emb = self.audio(ex[:, :use].reshape(b * use, 96, 64)).view(b, use, 128)
        if use < num_frames:  # compact_audio_feature: repeat the last row
            emb = torch.cat([emb, emb[:, -1:].expand(b, num_frames - use, 128)], dim=1)
        return emb.reshape(b, 1, num_frames, 128).contiguous()

    @staticmethod
    def frame_token_index(attention_mask_cpu, num_frames):
        """Host-side index plan: for every (clip, frame) the flat row of the token it shows, or -1.
        Applies exclude_padding (speech.py:567-586) and align_word_embedding_new (speech.py:690-738)."""
        bsz, s = attention_mask_cpu.shape
        plan = torch.full((bsz, num_frames), -1, dtype=torch.long)
        for b in range(bsz):
            attended = torch.nonzero(attention_mask_cpu[b] == 1).flatten().tolist()
            if len(attended) == s:
                raise ValueError("The sentence is too long, enlarge the token number!")
            words = attended[1:-1]  # drop [CLS] and the last attended token ([SEP])
            idx = align_tokens_to_frames(len(words), num_frames)
            for f, j in enumerate(idx):
                plan[b, f] = b * s + words[j]
        return plan

    @torch.no_grad()
    def text_features(self, token_ids, attention_mask, num_frames, attention_mask_cpu=None):
        """ids/mask [B,S] (one padded sentence per clip) -> [B,1,L,768].  Pass the loader's CPU copy of
        the mask as ``attention_mask_cpu`` to avoid a device->host copy."""
        tok = self.text(token_ids, attention_mask)
        bsz, s, hd = tok.shape
        mask_cpu = attention_mask_cpu if attention_mask_cpu is not None else attention_mask.cpu()
        plan = self.frame_token_index(mask_cpu, num_frames).to(tok.device)
        rows = ops.gather_rows(tok.view(bsz * s, hd), plan.view(-1).contiguous())  # frames without a token (-1) stay zero
        return rows.view(bsz, 1, num_frames, hd)

    @torch.no_grad()
    def forward(self, frames, pcm_int16, token_ids, attention_mask, attention_mask_cpu=None):
        """-> dict in the model's modality order {video, vggish, bert}."""
        length = frames.shape[1]
        return {"video": frames, "vggish": self.audio_features(pcm_int16, length),
                "bert": self.text_features(token_ids, attention_mask, length, attention_mask_cpu)}
