#!/usr/bin/env python
"""Known answers for the NLLB-200 / M2M-100 network from `transformers`' own implementation
(M2M100ForConditionalGeneration, transformers 5.15.0 in this image) - the published network behind the third-party `nllw`
package the reference loads for config 5 (whisperlivekit/core.py:320-329; `nllw` itself is not in the reference tree).

A micro configuration (whisperlivekit_amd.nllb.NLLB_MICRO: 2 + 2 layers, 2 heads of 64, FFN 256, 2003 tokens) gets the
seeded weights of whisperlivekit_amd.nllb.synth_state_dict (so tests regenerate them from the seed), then for a few source
sentences of ragged length:

* the encoder output,
* teacher-forced decoder logits for a fixed target prefix (prefill of all tokens at once, and token by token with the KV
  cache - the two must agree in the implementation under test),
* `model.generate(num_beams=1, do_sample=False, forced_bos_token_id=...)` with and without a forced `</s>` at the length limit.

Writes tests/golden/nllb_kat.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from transformers import M2M100Config, M2M100ForConditionalGeneration

    from whisperlivekit_amd import nllb

    cfg = nllb.NLLB_MICRO
    hf_cfg = M2M100Config(vocab_size=cfg.vocab_size, d_model=cfg.d_model, encoder_layers=cfg.encoder_layers,
                          decoder_layers=cfg.decoder_layers, encoder_attention_heads=cfg.attention_heads,
                          decoder_attention_heads=cfg.attention_heads, encoder_ffn_dim=cfg.ffn_dim, decoder_ffn_dim=cfg.ffn_dim,
                          activation_function="relu", scale_embedding=cfg.scale_embedding, pad_token_id=cfg.pad_token_id,
                          eos_token_id=cfg.eos_token_id, bos_token_id=0, decoder_start_token_id=cfg.decoder_start_token_id,
                          max_position_embeddings=cfg.max_position_embeddings, dropout=0.0, attention_dropout=0.0,
                          activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0, use_cache=True)
    model = M2M100ForConditionalGeneration(hf_cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in nllb.synth_state_dict(cfg, 0).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    tied = {"lm_head.weight", "model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight"}
    assert set(missing) <= tied and not unexpected, (missing, unexpected)
    model.tie_weights()
    assert torch.equal(model.lm_head.weight, sd["model.shared.weight"])
    assert torch.equal(model.model.encoder.embed_tokens.weight, sd["model.shared.weight"])

    rng = np.random.default_rng(7)
    out = {}
    cases = []
    for ci, (n_src, n_tgt, lang, max_new) in enumerate([(1, 3, 1990, 8), (5, 9, 1991, 24), (17, 6, 1992, 40), (64, 12, 1993, 30),
                                                          (90, 2, 1994, 12)]):
        src = rng.integers(4, 1900, size=n_src).astype(np.int64)
        src[-1] = cfg.eos_token_id                                   # NLLB sources end with </s> (and may contain it only there)
        tgt = np.concatenate([[cfg.decoder_start_token_id, lang], rng.integers(4, 1900, size=n_tgt)]).astype(np.int64)
        with torch.no_grad():
            ids = torch.from_numpy(src)[None]
            enc = model.model.encoder(input_ids=ids).last_hidden_state[0]
            logits = model(input_ids=ids, decoder_input_ids=torch.from_numpy(tgt)[None]).logits[0]
            gen = model.generate(ids, forced_bos_token_id=lang, num_beams=1, do_sample=False, max_new_tokens=max_new)[0]
            gen_eos = model.generate(ids, forced_bos_token_id=lang, forced_eos_token_id=cfg.eos_token_id, num_beams=1,
                                     do_sample=False, max_new_tokens=max_new)[0]
        out[f"src{ci}"], out[f"tgt{ci}"] = src, tgt
        out[f"enc{ci}"] = enc.numpy().astype(np.float32)
        out[f"logits{ci}"] = logits.numpy().astype(np.float32)
        out[f"gen{ci}"] = gen.numpy().astype(np.int64)
        out[f"gen_eos{ci}"] = gen_eos.numpy().astype(np.int64)
        cases.append((lang, max_new))
        print(f"case {ci}: src {n_src}, forced prefix {len(tgt)}, generate -> {gen.tolist()[:12]}... ({len(gen)} ids, "
              f"with forced eos {len(gen_eos)} ids, ends {int(gen_eos[-1])})")
    # ---- beam search: a second weight set whose </s> row is loud enough for hypotheses to end (seed 0, eos_gain 6), so that the
    # finished-slot bookkeeping, the length penalty and the three early-stopping modes all take part
    beam_model = M2M100ForConditionalGeneration(hf_cfg).eval()
    sd5 = {k: torch.from_numpy(v) for k, v in nllb.synth_state_dict(cfg, 0, eos_gain=6.0).items()}
    beam_model.load_state_dict(sd5, strict=False)
    beam_model.tie_weights()
    beam_cases = []
    specs = [(5, 3, 1.0, False, None, 24), (17, 3, 1.0, False, None, 24), (40, 3, 1.0, False, None, 24),
             (9, 4, 0.6, False, None, 30), (23, 2, 1.0, True, None, 30), (12, 4, 2.0, "never", None, 16),
             (31, 3, 1.0, False, cfg.eos_token_id, 12), (64, 5, 0.0, False, None, 20), (3, 8, 1.0, False, None, 10)]
    for bi, (n_src, beams, lp, es, feos, max_new) in enumerate(specs):
        src = rng.integers(4, 1900, size=n_src).astype(np.int64)
        src[-1] = cfg.eos_token_id
        kw = dict(forced_bos_token_id=1990 + bi, num_beams=beams, do_sample=False, max_new_tokens=max_new, length_penalty=lp,
                  early_stopping=es)
        if feos is not None:
            kw["forced_eos_token_id"] = feos
        with torch.no_grad():
            seq = beam_model.generate(torch.from_numpy(src)[None], **kw)[0]
        out[f"beam_src{bi}"] = src
        out[f"beam_out{bi}"] = seq.numpy().astype(np.int64)
        beam_cases.append((1990 + bi, beams, int(round(lp * 1000)), {False: 0, True: 1, "never": 2}[es], -1 if feos is None else feos, max_new))
        print(f"beam case {bi}: src {n_src}, {beams} beams, length_penalty {lp}, early_stopping {es}: {seq.tolist()}")
    out["beam_cases"] = np.asarray(beam_cases, np.int64)
    out["cases"] = np.asarray(cases, np.int64)
    path = os.path.join(ROOT, "tests", "golden", "nllb_kat.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
