#!/usr/bin/env python
"""Known answers for the Sortformer pass from the INDEPENDENT ports of NeMo's modules that ship with
`transformers` 5.15.0 in this image - NeMo itself (`nemo-toolkit[asr]`, pyproject.toml:80-85 of the reference) is not
installable here and no Sortformer checkpoint exists offline:

* `ParakeetFeatureExtractor` (transformers/models/parakeet/feature_extraction_parakeet.py) = NeMo's `FilterbankFeatures`
  as `AudioToMelSpectrogramPreprocessor` configures it (whisperlivekit/diarization/sortformer_backend.py:181-188):
  pre-emphasis 0.97, symmetric hann(400), centred zero-padded STFT(512, hop 160), power, slaney mel bank, log(x + 2^-24),
  valid frames = floor(len / hop) with the frames beyond them filled with 0.  Run with 128 bins.  Two views are stored:
  the un-normalised log-mel (`_torch_extract_fbank_features` on the pre-emphasised signal - normalize "NA" is what the
  reference asks for) and the extractor's own `__call__` output (per-feature normalisation + mask; the test normalises
  the features under test the same way).
* `ParakeetEncoder` (modeling_parakeet.py) = NeMo's FastConformer `ConformerEncoder`: dw-striding x8 sub-sampling stem,
  x sqrt(d) input scale, relative positional encoding, 17 x {FFN/2, rel-pos MHA with u/v biases + rel_shift, conv module
  (pointwise-GLU, depthwise 9, BatchNorm, SiLU, pointwise), FFN/2, LayerNorm} at the diar_streaming_sortformer_4spk-v2
  geometry (d 512, 8 heads, FFN 2048, 256 stem channels, 128 mel bins).  It gets the seeded weights of
  `whisperlivekit_amd.sortformer.synth_sortformer_state_dict` (NeMo parameter names -> HF names below), so the tests
  regenerate them from the seed.
* `BertEncoder` (modeling_bert.py, hidden_act relu, layer_norm_eps 1e-5, d 192, 8 heads, inner 768, 18 layers): NOT a
  NeMo port - an independent implementation of the same published post-LN Transformer encoder block NeMo's
  `TransformerEncoder(pre_ln=False)` builds for Sortformer.  It pins the block ARITHMETIC (softmax(QK^T / sqrt(dh)) V,
  residual -> LayerNorm, ReLU FFN, residual -> LayerNorm); that NeMo wires its block this way remains a restatement.

`librosa` (only `librosa.filters.mel`) and `soxr` are absent from the image: the extractor's filter bank comes from
`transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` - transformers' own float64 implementation
of the same bank - through a harness-side stub.

Writes tests/golden/sortformer_hf_kat.npz.  What stays unpinned afterwards: that NeMo's Transformer blocks / sigmoid
head are wired as restated, and the speaker-cache update (sortformer_backend.py:293-300).
"""
import importlib.machinery
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "sortformer_hf_kat.npz")

# the feature cases: (name, seconds of `synth.speech_like`, seed, cut to n samples or None)
FEATURE_CASES = [("chunk0", 1.0, 11, None), ("chunk1", 1.0, 12, None), ("half", 0.5, 13, None),
                 ("ragged", 1.0, 14, 12345), ("four_s", 4.0, 15, None), ("tiny", 0.1, 16, 1000)]
N_CTX = 95          # context rows in front of the 25 chunk rows for the [speaker cache | FIFO | chunk] case
TF_ROWS = 120


def install_stubs():
    """`librosa.filters.mel` over transformers' own slaney bank; `soxr` is imported next to librosa and never used."""
    def stub(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__version__ = "0.0.0"
        sys.modules[name] = m
        return m
    lib, filt = stub("librosa"), stub("librosa.filters")
    stub("soxr")

    def mel(sr, n_fft, n_mels, fmin=0.0, fmax=None, norm="slaney"):
        from transformers.audio_utils import mel_filter_bank
        m = mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
                            max_frequency=fmax if fmax is not None else sr / 2, sampling_rate=sr, norm=norm, mel_scale="slaney")
        return np.ascontiguousarray(m.T).astype(np.float32)
    filt.mel = mel
    lib.filters = filt


def nemo_to_hf_encoder(sd_nemo):
    """NeMo ConformerEncoder parameter names -> ParakeetEncoder's."""
    import torch
    attn = {"linear_q": "q_proj", "linear_k": "k_proj", "linear_v": "v_proj", "linear_out": "o_proj",
            "linear_pos": "relative_k_proj", "pos_bias_u": "bias_u", "pos_bias_v": "bias_v"}
    out = {}
    for k, v in sd_nemo.items():
        if not k.startswith("encoder."):
            continue
        k = k[len("encoder."):]
        if k.startswith("pre_encode.conv."):
            k = "subsampling.layers." + k[len("pre_encode.conv."):]
        elif k.startswith("pre_encode.out."):
            k = "subsampling.linear." + k[len("pre_encode.out."):]
        else:
            parts = k.split(".")
            if parts[2] == "self_attn":
                parts[3] = attn[parts[3]]
            if parts[2] == "conv" and parts[3] == "batch_norm":
                parts[3] = "norm"
            k = ".".join(parts)
        out[k] = torch.from_numpy(np.asarray(v))
    return out


def nemo_to_bert(sd_nemo, n_layers):
    import torch
    m = {"first_sub_layer.query_net": "attention.self.query", "first_sub_layer.key_net": "attention.self.key",
         "first_sub_layer.value_net": "attention.self.value", "first_sub_layer.out_projection": "attention.output.dense",
         "layer_norm_1": "attention.output.LayerNorm", "second_sub_layer.dense_in": "intermediate.dense",
         "second_sub_layer.dense_out": "output.dense", "layer_norm_2": "output.LayerNorm"}
    out = {}
    for i in range(n_layers):
        for src, dst in m.items():
            for leaf in ("weight", "bias"):
                out[f"layer.{i}.{dst}.{leaf}"] = torch.from_numpy(sd_nemo[f"transformer_encoder.layers.{i}.{src}.{leaf}"])
    return out


def main():
    install_stubs()
    import torch
    from transformers.models.bert.configuration_bert import BertConfig
    from transformers.models.bert.modeling_bert import BertEncoder
    from transformers.models.parakeet.configuration_parakeet import ParakeetEncoderConfig
    from transformers.models.parakeet.feature_extraction_parakeet import ParakeetFeatureExtractor
    from transformers.models.parakeet.modeling_parakeet import ParakeetEncoder

    from whisperlivekit_amd import synth
    from whisperlivekit_amd.sortformer import SortformerDims, synth_sortformer_state_dict

    torch.set_num_threads(8)
    out = {}
    # ---- (a) features ----------------------------------------------------------------------------------------
    fe = ParakeetFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, n_fft=512, win_length=400,
                                  preemphasis=0.97)
    out["mel_filters"] = fe.mel_filters.numpy()
    feats = {}
    for name, seconds, seed, cut in FEATURE_CASES:
        pcm = synth.speech_like(seconds, seed).astype(np.float32)
        if cut is not None:
            pcm = pcm[:cut]
        x = torch.from_numpy(pcm)[None]
        pre = torch.cat([x[:, :1], x[:, 1:] - fe.preemphasis * x[:, :-1]], dim=1)       # __call__'s pre-emphasis lines
        raw = fe._torch_extract_fbank_features(pre, "cpu")[0]                           # [frames, 128], normalize "NA"
        full = fe(pcm, sampling_rate=16000, return_tensors="pt")
        valid = int(full["attention_mask"][0].sum())
        assert valid == len(pcm) // 160 and raw.shape[0] == len(pcm) // 160 + 1
        out[f"feat_{name}_raw"] = raw.numpy()
        out[f"feat_{name}_normalized"] = full["input_features"][0].numpy()
        out[f"feat_{name}_valid"] = np.int64(valid)
        out[f"feat_{name}_n"] = np.int64(len(pcm))
        feats[name] = raw.numpy()
    # ---- (b) FastConformer ---------------------------------------------------------------------------------------
    dims = SortformerDims()
    sd = synth_sortformer_state_dict(dims, 0)
    cfg = ParakeetEncoderConfig(hidden_size=dims.fc_d_model, num_hidden_layers=dims.fc_layers,
                                num_attention_heads=dims.fc_heads, intermediate_size=dims.fc_ff, hidden_act="silu",
                                attention_bias=True, convolution_bias=True, conv_kernel_size=dims.conv_kernel,
                                subsampling_factor=8, subsampling_conv_channels=dims.sub_channels, num_mel_bins=dims.n_mels,
                                subsampling_conv_kernel_size=3, subsampling_conv_stride=2, dropout=0.0, dropout_positions=0.0,
                                layerdrop=0.0, activation_dropout=0.0, attention_dropout=0.0, scale_input=True)
    cfg._attn_implementation = "eager"
    enc = ParakeetEncoder(cfg).eval()
    missing, unexpected = enc.load_state_dict(nemo_to_hf_encoder(sd), strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
    # the diarizer's second chunk: the last 99 feature frames of chunk 0 in front of chunk 1's 101 (the zeroed last frame
    # of each chunk included, as the reference concatenates them: sortformer_backend.py:279-283)
    def masked(name):
        f = feats[name].copy()
        f[int(out[f"feat_{name}_valid"]):] = 0.0
        return f
    total = np.concatenate([masked("chunk0")[-99:], masked("chunk1")], axis=0)
    out["stem_in"] = total
    with torch.no_grad():
        x = torch.from_numpy(total)[None]
        stem = enc.subsampling(x, None)                                                       # [1, 25, 512]
        out["stem_out"] = stem[0].numpy()
        h = stem * enc.input_scale
        pos = enc.encode_positions(h)
        out["block0_out"] = enc.layers[0](h, attention_mask=None, position_embeddings=pos)[0].numpy()
        out["stack_chunk_out"] = enc(x).last_hidden_state[0].numpy()                          # features -> 17 blocks
        # [speaker cache | FIFO | chunk]: context embeddings in front of the chunk's (bypass_pre_encode=True in NeMo)
        rng = np.random.default_rng(5)
        ctx = (rng.standard_normal((N_CTX, dims.fc_d_model)) * float(stem.std())).astype(np.float32)
        out["ctx_embs"] = ctx
        h = torch.cat([torch.from_numpy(ctx)[None], stem], dim=1) * enc.input_scale
        pos = enc.encode_positions(h)
        for layer in enc.layers:
            h = layer(h, attention_mask=None, position_embeddings=pos)
        out["stack_ctx_out"] = h[0].numpy()
    # ---- (c) post-LN Transformer blocks (independent implementation, not a NeMo port) -------------------------------
    bcfg = BertConfig(hidden_size=dims.tf_d_model, num_hidden_layers=dims.tf_layers, num_attention_heads=dims.tf_heads,
                      intermediate_size=dims.tf_inner, hidden_act="relu", hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5)
    bcfg._attn_implementation = "eager"
    bert = BertEncoder(bcfg).eval()
    missing, unexpected = bert.load_state_dict(nemo_to_bert(sd, dims.tf_layers), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    rng = np.random.default_rng(6)
    tf_in = rng.standard_normal((TF_ROWS, dims.tf_d_model)).astype(np.float32)
    out["tf_in"] = tf_in
    with torch.no_grad():
        h = torch.from_numpy(tf_in)[None]
        r = bert.layer[0](h)
        out["tf_block0_out"] = (r[0] if isinstance(r, tuple) else r)[0].numpy()
        out["tf_stack_out"] = bert(h).last_hidden_state[0].numpy()
    np.savez_compressed(GOLDEN, **out)
    print(f"wrote {GOLDEN}: {os.path.getsize(GOLDEN) / 1e6:.2f} MB, {len(out)} arrays")
    for k in ("stem_out", "block0_out", "stack_chunk_out", "stack_ctx_out", "tf_block0_out", "tf_stack_out"):
        print(k, out[k].shape, float(np.abs(out[k]).mean()))


if __name__ == "__main__":
    main()
