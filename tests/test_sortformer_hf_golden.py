"""a12 parity pin (round 5): the Sortformer oracle against known answers from the INDEPENDENT ports of NeMo's modules in
`transformers` 5.15.0 - `ParakeetFeatureExtractor` (NeMo FilterbankFeatures), `ParakeetEncoder` (NeMo FastConformer) - and
against `BertEncoder`'s post-LN block for the Transformer part (tests/golden/sortformer_hf_kat.npz, made by
scripts/gen_golden_sortformer_hf.py with the seeded weights of `sortformer.synth_sortformer_state_dict(SortformerDims(), 0)`).
The HIP path is compared with the same file in tests/test_gpu_sortformer.py.  Tolerances are absolute on O(1) values."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import sortformer_oracle as so
from whisperlivekit_amd import sortformer as sf
from whisperlivekit_amd import synth
from whisperlivekit_amd.melbank import mel_filterbank

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sortformer_hf_kat.npz")
FEATURE_CASES = [("chunk0", 1.0, 11, None), ("chunk1", 1.0, 12, None), ("half", 0.5, 13, None),
                 ("ragged", 1.0, 14, 12345), ("four_s", 4.0, 15, None), ("tiny", 0.1, 16, 1000)]


def feature_case_pcm(seconds, seed, cut):
    pcm = synth.speech_like(seconds, seed).astype(np.float32)
    return pcm if cut is None else pcm[:cut]


def per_feature_normalized(feats, valid):
    """ParakeetFeatureExtractor.__call__'s tail (NeMo normalize='per_feature'): mean / unbiased std over the valid
    frames, frames beyond them zero."""
    x = feats[:valid].astype(np.float64)
    mean = x.mean(axis=0)
    std = np.sqrt(((x - mean) ** 2).sum(axis=0) / (valid - 1))
    out = np.zeros_like(feats, dtype=np.float64)
    out[:valid] = (x - mean) / (std + 1e-5)
    return out


@pytest.fixture(scope="module")
def kat():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def weights():
    dims = sf.SortformerDims()
    sd = {k: torch.from_numpy(v) for k, v in sf.synth_sortformer_state_dict(dims, 0).items()}
    od = so.SortformerDims(n_mels=dims.n_mels, fc_d_model=dims.fc_d_model, fc_layers=dims.fc_layers, fc_heads=dims.fc_heads,
                           conv_kernel=dims.conv_kernel, sub_channels=dims.sub_channels, tf_d_model=dims.tf_d_model,
                           tf_layers=dims.tf_layers, tf_heads=dims.tf_heads, tf_inner=dims.tf_inner, n_spk=dims.n_spk)
    return sd, od


def test_mel_filterbank_is_transformers_slaney_bank(kat):
    ours = np.asarray(mel_filterbank(128, 16000, 512), np.float32)
    assert ours.shape == kat["mel_filters"].shape == (128, 257)
    assert np.abs(ours - kat["mel_filters"]).max() <= 1e-8


@pytest.mark.parametrize("name,seconds,seed,cut", FEATURE_CASES)
def test_log_mel_matches_parakeet_feature_extractor(kat, name, seconds, seed, cut):
    pcm = feature_case_pcm(seconds, seed, cut)
    assert len(pcm) == int(kat[f"feat_{name}_n"])
    valid = int(kat[f"feat_{name}_valid"])
    assert valid == len(pcm) // 160
    filters = np.asarray(mel_filterbank(128, 16000, 512), np.float32)
    got = so.nemo_log_mel(pcm, filters)
    raw = kat[f"feat_{name}_raw"]
    assert got.shape == raw.shape == (len(pcm) // 160 + 1, 128)
    # valid frames: the log-mel; the frame(s) behind them: pad_value 0 (the extractor's `input_features *= mask`)
    assert np.abs(got[:valid] - raw[:valid]).max() <= 2e-5
    assert np.all(got[valid:] == 0.0) and np.all(kat[f"feat_{name}_normalized"][valid:] == 0.0)
    # the extractor's own end-to-end output (pre-emphasis, padding and masking through ITS code path)
    if valid > 1:
        norm = per_feature_normalized(got, valid)
        assert np.abs(norm - kat[f"feat_{name}_normalized"]).max() <= 2e-4
    # the pre-2.0 rule keeps the last frame
    old = so.nemo_log_mel(pcm, filters, seq_len_plus_one=True)
    assert np.abs(old - raw).max() <= 2e-5


def test_subsampling_stem_matches_parakeet_encoder(kat, weights):
    sd, od = weights
    with torch.no_grad():
        got = so.pre_encode(sd, od, torch.from_numpy(kat["stem_in"])).numpy()
    assert got.shape == kat["stem_out"].shape == (25, 512)
    assert np.abs(got - kat["stem_out"]).max() <= 1e-5


def test_conformer_block_and_stack_match_parakeet_encoder(kat, weights):
    sd, od = weights
    with torch.no_grad():
        stem = torch.from_numpy(kat["stem_out"])
        x = stem * math.sqrt(od.fc_d_model)
        pos = so.rel_positional_encoding(x.shape[0], od.fc_d_model)
        b0 = so.conformer_layer(sd, "encoder.layers.0.", od, x, pos).numpy()
        assert np.abs(b0 - kat["block0_out"]).max() <= 1e-5
        stack = so.conformer_stack(sd, od, stem).numpy()
        assert np.abs(stack - kat["stack_chunk_out"]).max() <= 1e-5
        embs = torch.cat([torch.from_numpy(kat["ctx_embs"]), stem], 0)
        ctx = so.conformer_stack(sd, od, embs).numpy()
        assert ctx.shape == kat["stack_ctx_out"].shape == (120, 512)
        assert np.abs(ctx - kat["stack_ctx_out"]).max() <= 1e-5


def test_transformer_block_and_stack_match_bert_post_ln_encoder(kat, weights):
    sd, od = weights
    with torch.no_grad():
        y = so.transformer_layer(sd, "transformer_encoder.layers.0.", od, torch.from_numpy(kat["tf_in"]))
        assert np.abs(y.numpy() - kat["tf_block0_out"]).max() <= 1e-5
        for i in range(1, od.tf_layers):
            y = so.transformer_layer(sd, f"transformer_encoder.layers.{i}.", od, y)
        assert np.abs(y.numpy() - kat["tf_stack_out"]).max() <= 2e-5
