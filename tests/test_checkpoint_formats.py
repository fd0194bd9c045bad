"""a15 beyond the `.pt` branch (round 5): everything the reference's load_model takes as a LOCAL checkpoint
(whisper/__init__.py:163-271, 337-596; model_paths.py:69-177) through `whisperlivekit_amd.checkpoint` - HuggingFace names,
safetensors, shards with and without an index, MLX names with their own alignment heads, a PEFT LoRA adapter merged in,
dimensions from config.json.  The HuggingFace checkpoints are REAL ones: `transformers`' own
`WhisperForConditionalGeneration` at micro dimensions, saved by `save_pretrained`.  Where the reference tree is present the
reference's own `load_model` reads the same directories and must end up with the same dimensions and tensors; on the GPU
the HIP model built from such a directory is compared with `transformers`' forward pass - an implementation of the network
that shares no code with the reference or this repository."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import helpers as H
from whisperlivekit_amd import checkpoint as ck
from whisperlivekit_amd.dims import MODEL_DIMS, ModelDims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402

MICRO = MODEL_DIMS["micro.en"]


def hf_micro_model(seed=0):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    d = MICRO
    cfg = WhisperConfig(vocab_size=d.n_vocab, num_mel_bins=d.n_mels, d_model=d.n_audio_state, encoder_layers=d.n_audio_layer,
                        decoder_layers=d.n_text_layer, encoder_attention_heads=d.n_audio_head, decoder_attention_heads=d.n_text_head,
                        encoder_ffn_dim=4 * d.n_audio_state, decoder_ffn_dim=4 * d.n_text_state, max_source_positions=d.n_audio_ctx,
                        max_target_positions=d.n_text_ctx, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                        activation_function="gelu", pad_token_id=50256, bos_token_id=50256, eos_token_id=50256,
                        decoder_start_token_id=50257, suppress_tokens=None, begin_suppress_tokens=None)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    model = WhisperForConditionalGeneration(cfg).eval()
    with torch.no_grad():                  # default initialisation leaves biases at zero and LayerNorms at identity: perturb them
        g = torch.Generator().manual_seed(seed + 1)
        for n, p in model.named_parameters():
            if n.endswith(".bias") or "layer_norm" in n:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return model


@pytest.fixture(scope="module")
def hf_dirs(tmp_path_factory):
    """One micro checkpoint saved four ways by transformers / torch: single safetensors, shards + index, shards without the
    index, a bare `pytorch_model.bin`."""
    model = hf_micro_model()
    base = tmp_path_factory.mktemp("hf")
    single, sharded, noindex, binary = (str(base / n) for n in ("single", "sharded", "noindex", "bin"))
    model.save_pretrained(single, safe_serialization=True)
    model.save_pretrained(sharded, safe_serialization=True, max_shard_size="6MB")
    assert os.path.isfile(os.path.join(sharded, "model.safetensors.index.json")), os.listdir(sharded)
    model.save_pretrained(noindex, safe_serialization=True, max_shard_size="6MB")
    os.remove(os.path.join(noindex, "model.safetensors.index.json"))
    os.makedirs(binary)
    torch.save(model.state_dict(), os.path.join(binary, "pytorch_model.bin"))
    with open(os.path.join(binary, "config.json"), "w") as fh:
        json.dump(model.config.to_dict(), fh)
    return dict(model=model, single=single, sharded=sharded, noindex=noindex, bin=binary)


def expected_openai_sd(model):
    """The mapping written out by hand for the tensors of a 2 + 2 layer model (independent of checkpoint.py's tables)."""
    sd = model.state_dict()
    out = {}
    for side, n in (("encoder", MICRO.n_audio_layer), ("decoder", MICRO.n_text_layer)):
        for i in range(n):
            src, dst = f"model.{side}.layers.{i}", f"{side}.blocks.{i}"
            pairs = [("self_attn.q_proj", "attn.query"), ("self_attn.k_proj", "attn.key"), ("self_attn.v_proj", "attn.value"),
                     ("self_attn.out_proj", "attn.out"), ("self_attn_layer_norm", "attn_ln"), ("fc1", "mlp.0"), ("fc2", "mlp.2"),
                     ("final_layer_norm", "mlp_ln")]
            if side == "decoder":
                pairs += [("encoder_attn.q_proj", "cross_attn.query"), ("encoder_attn.k_proj", "cross_attn.key"),
                          ("encoder_attn.v_proj", "cross_attn.value"), ("encoder_attn.out_proj", "cross_attn.out"),
                          ("encoder_attn_layer_norm", "cross_attn_ln")]
            for a, b in pairs:
                for leaf in ("weight", "bias"):
                    if f"{src}.{a}.{leaf}" in sd:
                        out[f"{dst}.{b}.{leaf}"] = sd[f"{src}.{a}.{leaf}"]
    for c in ("conv1", "conv2"):
        for leaf in ("weight", "bias"):
            out[f"encoder.{c}.{leaf}"] = sd[f"model.encoder.{c}.{leaf}"]
    out["encoder.positional_embedding"] = sd["model.encoder.embed_positions.weight"]
    out["decoder.positional_embedding"] = sd["model.decoder.embed_positions.weight"]
    out["decoder.token_embedding.weight"] = sd["model.decoder.embed_tokens.weight"]
    for leaf in ("weight", "bias"):
        out[f"encoder.ln_post.{leaf}"] = sd[f"model.encoder.layer_norm.{leaf}"]
        out[f"decoder.ln.{leaf}"] = sd[f"model.decoder.layer_norm.{leaf}"]
    return out


@pytest.mark.parametrize("kind", ["single", "sharded", "noindex", "bin"])
def test_huggingface_checkpoints_are_read_back(hf_dirs, kind):
    dims, sd, heads = ck.load_whisper_checkpoint(hf_dirs[kind])
    assert dims == MICRO and heads is None
    want = expected_openai_sd(hf_dirs["model"])
    assert set(sd) == set(want) == set(H.synth_sd("micro.en"))            # exactly the openai parameter names
    assert all(torch.equal(torch.as_tensor(sd[k]), want[k]) for k in want)
    if kind in ("sharded", "noindex"):
        assert len(ck.collect_checkpoint_files(hf_dirs[kind])) > 1
    # a single FILE of the directory works as well (dims from the config.json beside it)
    if kind == "single":
        d2, sd2, _ = ck.load_whisper_checkpoint(os.path.join(hf_dirs[kind], "model.safetensors"))
        assert d2 == MICRO and set(sd2) == set(want)


@pytest.mark.skipif(not ref_stubs.reference_available(), reason="reference tree not present")
@pytest.mark.parametrize("kind", ["single", "sharded", "noindex", "bin"])
def test_same_tensors_as_the_references_load_model(hf_dirs, kind):
    ref_stubs.install(synthetic_vocab=True)
    from whisperlivekit import whisper as ref_whisper
    ref = ref_whisper.load_model(hf_dirs[kind], device="cpu")
    dims, sd, _ = ck.load_whisper_checkpoint(hf_dirs[kind])
    assert dims.as_tuple() == tuple(getattr(ref.dims, f) for f in ck._NATIVE)
    ref_sd = ref.state_dict()
    assert set(ref_sd) == set(sd)
    assert all(torch.equal(ref_sd[k], torch.as_tensor(sd[k])) for k in sd)


def test_mlx_names_and_their_alignment_heads(tmp_path):
    sd = {k.replace(".mlp.0.", ".mlp1.").replace(".mlp.2.", ".mlp2."): torch.from_numpy(v) for k, v in H.synth_sd("micro.en").items()}
    sd["alignment_heads"] = torch.tensor([[1, 0], [1, 1]])
    from safetensors.torch import save_file
    save_file(sd, str(tmp_path / "weights.safetensors"))
    with open(tmp_path / "config.json", "w") as fh:
        json.dump({k: getattr(MICRO, k) for k in ck._NATIVE}, fh)            # the native-key config MLX exports carry
    dims, got, heads = ck.load_whisper_checkpoint(str(tmp_path))
    assert dims == MICRO and heads == [(1, 0), (1, 1)]
    want = H.synth_sd("micro.en")
    assert set(got) == set(want) and all(np.array_equal(np.asarray(got[k]), want[k]) for k in want)


def lora_dir(tmp_path, model, as_bin=False, r=4, alpha=8.0):
    d = MICRO.n_text_state
    g = torch.Generator().manual_seed(5)
    adapter = {}
    for mod in ("decoder.layers.0.self_attn.q_proj", "decoder.layers.1.encoder_attn.v_proj", "encoder.layers.1.self_attn.out_proj",
                "decoder.layers.1.fc1"):
        out_f = model.state_dict()[f"model.{mod}.weight"].shape[0]
        adapter[f"base_model.model.model.{mod}.lora_A.weight"] = 0.1 * torch.randn(r, d, generator=g)
        adapter[f"base_model.model.model.{mod}.lora_B.weight"] = 0.1 * torch.randn(out_f, r, generator=g)
    p = tmp_path / ("lora_bin" if as_bin else "lora")
    p.mkdir()
    if as_bin:
        torch.save(adapter, str(p / "adapter_model.bin"))
    else:
        from safetensors.torch import save_file
        save_file(adapter, str(p / "adapter_model.safetensors"))
    with open(p / "adapter_config.json", "w") as fh:
        json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha}, fh)
    return str(p), adapter, alpha / r


@pytest.mark.parametrize("as_bin", [False, True])
def test_lora_adapter_is_merged(hf_dirs, tmp_path, as_bin):
    path, adapter, scaling = lora_dir(tmp_path, hf_dirs["model"], as_bin)
    _, base, _ = ck.load_whisper_checkpoint(hf_dirs["single"])
    dims, merged, _ = ck.load_whisper_checkpoint(hf_dirs["single"], lora_path=path)
    touched = {"decoder.blocks.0.attn.query.weight": "decoder.layers.0.self_attn.q_proj",
               "decoder.blocks.1.cross_attn.value.weight": "decoder.layers.1.encoder_attn.v_proj",
               "encoder.blocks.1.attn.out.weight": "encoder.layers.1.self_attn.out_proj",
               "decoder.blocks.1.mlp.0.weight": "decoder.layers.1.fc1"}
    for k in base:
        if k in touched:
            a = adapter[f"base_model.model.model.{touched[k]}.lora_A.weight"]
            b = adapter[f"base_model.model.model.{touched[k]}.lora_B.weight"]
            assert torch.equal(torch.as_tensor(merged[k]), base[k] + (b @ a) * scaling)
            assert not torch.equal(torch.as_tensor(merged[k]), base[k])
        else:
            assert torch.equal(torch.as_tensor(merged[k]), torch.as_tensor(base[k]))
    if ref_stubs.reference_available():
        ref_stubs.install(synthetic_vocab=True)
        from whisperlivekit import whisper as ref_whisper
        ref_sd = ref_whisper.load_model(hf_dirs["single"], device="cpu", lora_path=path).state_dict()
        assert all(torch.equal(ref_sd[k], torch.as_tensor(merged[k])) for k in merged)


def test_lora_errors(hf_dirs, tmp_path):
    path, _, _ = lora_dir(tmp_path, hf_dirs["model"])
    with open(os.path.join(path, "adapter_config.json"), "w") as fh:
        json.dump({"peft_type": "IA3", "r": 4, "lora_alpha": 8}, fh)
    with pytest.raises(ValueError):
        ck.load_whisper_checkpoint(hf_dirs["single"], lora_path=path)
    os.remove(os.path.join(path, "adapter_config.json"))
    with pytest.raises(FileNotFoundError):
        ck.load_whisper_checkpoint(hf_dirs["single"], lora_path=path)
    from safetensors.torch import save_file
    bad = tmp_path / "bad"
    bad.mkdir()
    save_file({"base_model.model.model.decoder.layers.7.self_attn.q_proj.lora_A.weight": torch.zeros(4, 128),
               "base_model.model.model.decoder.layers.7.self_attn.q_proj.lora_B.weight": torch.zeros(128, 4)},
              str(bad / "adapter_model.safetensors"))
    with open(bad / "adapter_config.json", "w") as fh:
        json.dump({"peft_type": "LORA", "r": 4, "lora_alpha": 8}, fh)
    with pytest.raises(KeyError):                                # layer 7 does not exist in a 2-layer model
        ck.load_whisper_checkpoint(hf_dirs["single"], lora_path=str(bad))


def test_file_selection_rules(tmp_path):
    def touch(d, *names):
        os.makedirs(d, exist_ok=True)
        for n in names:
            open(os.path.join(d, n), "wb").close()
        return d
    base = str(tmp_path)
    assert ck.collect_checkpoint_files(touch(base + "/a", "x.pt", "pytorch_model.bin", "model.safetensors", "other.safetensors")) == \
        [base + "/a/model.safetensors"]
    assert ck.collect_checkpoint_files(touch(base + "/b", "x.pt", "pytorch_model.bin", "notes.txt")) == [base + "/b/pytorch_model.bin"]
    assert ck.collect_checkpoint_files(touch(base + "/c", "adapter_model.safetensors", "w.safetensors", "y.pt")) == [base + "/c/y.pt"]
    assert ck.collect_checkpoint_files(touch(base + "/d", "adapter_model.safetensors", "adapter_config.json")) == []
    # a complete shard group wins over single files; an incomplete one is ignored
    got = ck.collect_checkpoint_files(touch(base + "/e", "model-00002-of-00002.safetensors", "model-00001-of-00002.safetensors", "z.pt"))
    assert got == [base + "/e/model-00001-of-00002.safetensors", base + "/e/model-00002-of-00002.safetensors"]
    assert ck.collect_checkpoint_files(touch(base + "/f", "pytorch_model-00001-of-00003.bin", "z.pt")) == [base + "/f/z.pt"]
    # an index names the shards (only those that exist)
    d = touch(base + "/g", "s1.safetensors", "s2.safetensors", "model.safetensors")
    with open(os.path.join(d, "model.safetensors.index.json"), "w") as fh:
        json.dump({"weight_map": {"a": "s2.safetensors", "b": "s1.safetensors", "c": "s1.safetensors", "d": "gone.safetensors"}}, fh)
    assert ck.collect_checkpoint_files(d) == [d + "/s1.safetensors", d + "/s2.safetensors"]
    with pytest.raises(RuntimeError):
        ck.load_whisper_checkpoint(base + "/d")
    with pytest.raises(RuntimeError):
        ck.load_whisper_checkpoint(base + "/nowhere")
    if ref_stubs.reference_available():
        ref_stubs.install(synthetic_vocab=True)
        from whisperlivekit.model_paths import detect_model_format
        for sub in "abcdefg":
            assert [str(p) for p in detect_model_format(f"{base}/{sub}").pytorch_files] == ck.collect_checkpoint_files(f"{base}/{sub}"), sub


def test_dimensions_from_config_json(tmp_path):
    hf = {"num_mel_bins": 128, "max_source_positions": 1500, "d_model": 1280, "encoder_attention_heads": 20, "encoder_layers": 32,
          "vocab_size": 51866, "max_target_positions": 448, "decoder_attention_heads": 20, "decoder_layers": 4}     # a distilled model
    assert ck.dims_from_config(hf) == ModelDims(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4)
    hf2 = dict(hf)
    del hf2["encoder_layers"]
    hf2["num_hidden_layers"] = 12
    assert ck.dims_from_config(hf2).n_audio_layer == 12
    assert ck.dims_from_config({"d_model": 4}) is None
    from safetensors.torch import save_file
    save_file({k: torch.from_numpy(v) for k, v in H.synth_sd("micro.en").items()}, str(tmp_path / "model.safetensors"))
    with pytest.raises(RuntimeError, match="dimensions"):
        ck.load_whisper_checkpoint(str(tmp_path))                  # openai names, no dims, no config.json
    assert ck.hf_to_openai_name("proj_out.weight") is None and ck.hf_to_openai_name("model.decoder.layers.0.unknown.weight") is None


def _hf_reference_run(model, mel, feeds):
    with torch.no_grad():
        enc = model.model.encoder(input_features=torch.from_numpy(mel)[None]).last_hidden_state
        rows, past = [], None
        for toks in feeds:
            r = model(encoder_outputs=(enc,), decoder_input_ids=torch.from_numpy(toks), past_key_values=past, use_cache=True)
            past = r.past_key_values
            rows.append(r.logits[0, -1].numpy())
    return enc[0].numpy(), rows


def test_oracle_on_the_converted_tensors_matches_transformers_forward(hf_dirs):
    """An independent pin of the ORACLE's network (a3 / a5: it restates the reference's model.py): on the tensors
    `checkpoint.py` hands over it must reproduce `transformers`' own Whisper forward pass - encoder output and three cached
    decoder steps (HF scales q by d^-0.5, the reference q and k by d^-0.25: same product)."""
    from oracle import whisper_oracle as wo
    from whisperlivekit_amd import synth
    dims, sd, _ = ck.load_whisper_checkpoint(hf_dirs["single"])
    sd = {k: torch.as_tensor(v).float() for k, v in sd.items()}
    torch.set_num_threads(8)
    mel = wo.log_mel_spectrogram(torch.from_numpy(synth.speech_like(30.0, 3)), torch.from_numpy(np.array(H.mel_filterbank(80))))[:, :3000]
    mel = mel.contiguous().numpy().astype(np.float32)
    feeds = [np.asarray([[50257, 50362, 11, 250, 9999]], np.int64), np.asarray([[31000]], np.int64), np.asarray([[46]], np.int64)]
    want_enc, want_rows = _hf_reference_run(hf_dirs["model"], mel, feeds)
    with torch.no_grad():
        enc = wo.encoder_forward(sd, dims, torch.from_numpy(mel)[None])
        assert float((enc[0] - torch.from_numpy(want_enc)).abs().max()) <= 2e-4
        cache = wo.DecoderCache(dims.n_text_layer)
        for i, toks in enumerate(feeds):
            logits, _ = wo.decoder_forward(sd, dims, torch.from_numpy(toks), enc, cache)
            err = float((logits[0, -1] - torch.from_numpy(want_rows[i])).abs().max())
            assert err <= 5e-4, (i, err)
            assert int(logits[0, -1].argmax()) == int(want_rows[i].argmax())


@pytest.mark.gpu
def test_hip_model_from_a_huggingface_directory_matches_transformers_forward(hf_dirs):
    """`HipSimulStreamingASR(model_path=<HF directory>)`: encoder output and three cached decoder steps of the HIP library
    against `transformers`' WhisperForConditionalGeneration on the same weights and the same log-mel segment."""
    from whisperlivekit_amd import synth
    from whisperlivekit_amd.backend import HipSimulStreamingASR
    asr = HipSimulStreamingASR("micro.en", model_path=hf_dirs["sharded"])
    sess = asr.hip_model.new_session()
    try:
        mel = sess.log_mel(synth.speech_like(30.0, 3))[:, :3000].copy()
        feeds = [np.asarray([[50257, 50362, 11, 250, 9999]], np.int64), np.asarray([[31000]], np.int64), np.asarray([[46]], np.int64)]
        want_enc, want_rows = _hf_reference_run(hf_dirs["model"], mel, feeds)
        sess.encode_mel(mel)
        enc = sess.export("enc").reshape(MICRO.n_audio_ctx, MICRO.n_audio_state)
        assert float(np.abs(enc - want_enc).max()) <= 1e-3, float(np.abs(enc - want_enc).max())
        for i, toks in enumerate(feeds):
            sess.decode(toks, first=(i == 0), sot_index=0)
            got = sess.export("logits_last").reshape(-1)
            err = float(np.abs(got - want_rows[i]).max())
            assert err <= 1e-3, (i, err)
            assert int(got.argmax()) == int(want_rows[i].argmax()) or abs(want_rows[i][int(got.argmax())] - want_rows[i].max()) < 1e-4
    finally:
        sess.close()
