"""NLLB-200 / M2M-100 translation network: oracle and HIP library against `transformers`' own
M2M100ForConditionalGeneration on seeded weights (tests/golden/nllb_kat.npz, scripts/gen_golden_nllb.py)."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle.nllb_oracle import NllbOracle, OracleNllbSession
from whisperlivekit_amd import nllb

KAT = H.golden_npz("nllb_kat.npz")
N_CASES = len(KAT["cases"])
ENC_ATOL, LOGIT_ATOL = 2e-4, 1e-3      # fp32 accumulation-order differences; logits span about +-3


@pytest.fixture(scope="module")
def micro_oracle():
    return NllbOracle(nllb.NLLB_MICRO, nllb.synth_state_dict(nllb.NLLB_MICRO, 0))


def test_sinusoid_table_is_the_published_one():
    """Rows 0 / padding / a few positions of the table against closed forms (tensor2tensor layout: sines then cosines)."""
    t = nllb.sinusoid_table(40, 128, 1)
    assert t.shape == (40, 128) and not t[1].any()
    np.testing.assert_allclose(t[0, :64], 0.0, atol=0)
    np.testing.assert_allclose(t[0, 64:], 1.0, atol=0)
    np.testing.assert_allclose(t[5, 0], np.sin(5.0), rtol=1e-6)
    np.testing.assert_allclose(t[7, 64 + 63], np.cos(7.0 * 1e-4), rtol=1e-6)


@pytest.mark.parametrize("ci", range(N_CASES))
def test_oracle_matches_transformers(micro_oracle, ci):
    src, tgt = KAT[f"src{ci}"], KAT[f"tgt{ci}"]
    enc = micro_oracle.encode(src)
    np.testing.assert_allclose(enc.numpy(), KAT[f"enc{ci}"], rtol=0, atol=2e-5)
    cache = micro_oracle.new_cache()
    logits = micro_oracle.decode(torch.from_numpy(tgt)[None], enc, cache)[0]
    np.testing.assert_allclose(logits.numpy(), KAT[f"logits{ci}"], rtol=0, atol=5e-5)
    cache = micro_oracle.new_cache()                                   # token by token over the KV cache
    for i, t in enumerate(tgt):
        step = micro_oracle.decode(torch.tensor([[int(t)]]), enc, cache)[0, 0]
        np.testing.assert_allclose(step.numpy(), KAT[f"logits{ci}"][i], rtol=0, atol=5e-5)


@pytest.mark.parametrize("ci", range(N_CASES))
def test_generate_over_the_oracle_matches_transformers(micro_oracle, ci):
    lang, max_new = (int(v) for v in KAT["cases"][ci])
    sess = OracleNllbSession(micro_oracle, 1)
    assert nllb.generate(sess, KAT[f"src{ci}"], lang, max_new_tokens=max_new) == KAT[f"gen{ci}"].tolist()
    assert nllb.generate(sess, KAT[f"src{ci}"], lang, max_new_tokens=max_new,
                         forced_eos_token_id=nllb.NLLB_MICRO.eos_token_id) == KAT[f"gen_eos{ci}"].tolist()


N_BEAM = len(KAT["beam_cases"])
BEAM_WEIGHTS = dict(seed=0, eos_gain=6.0)       # scripts/gen_golden_nllb.py: a </s> row loud enough for hypotheses to end


def _beam_args(bi):
    lang, beams, lp1000, es, feos, max_new = (int(v) for v in KAT["beam_cases"][bi])
    return lang, dict(num_beams=beams, max_new_tokens=max_new, length_penalty=lp1000 / 1000.0,
                      early_stopping={0: False, 1: True, 2: "never"}[es], forced_eos_token_id=None if feos < 0 else feos)


@pytest.fixture(scope="module")
def beam_oracle():
    return NllbOracle(nllb.NLLB_MICRO, nllb.synth_state_dict(nllb.NLLB_MICRO, BEAM_WEIGHTS["seed"], BEAM_WEIGHTS["eos_gain"]))


@pytest.mark.parametrize("bi", range(N_BEAM))
def test_beam_search_over_the_oracle_matches_transformers(beam_oracle, bi):
    """whisperlivekit_amd.nllb.beam_search against transformers' generate(num_beams = 2 .. 8): hypotheses that end early
    and ones that run into the length limit, length penalties 0 / 0.6 / 1 / 2, the three early-stopping modes, a forced
    last token."""
    lang, kw = _beam_args(bi)
    sess = OracleNllbSession(beam_oracle, kw["num_beams"])
    assert nllb.beam_search(sess, KAT[f"beam_src{bi}"], lang, **kw) == KAT[f"beam_out{bi}"].tolist()


def test_pack_names_cover_the_arena_layout():
    """Every packed tensor the library expects is produced by pack_hf_state_dict, with the right size (no GPU needed)."""
    import ctypes as C
    from whisperlivekit_amd import _lib
    cfg = nllb.NLLB_MICRO
    lib = _lib.load()
    dims = _lib.NllbDims(cfg.vocab_size, cfg.d_model, cfg.attention_heads, cfg.ffn_dim, cfg.encoder_layers, cfg.decoder_layers,
                         64, 64, cfg.pad_token_id, 64 + 3, float(np.sqrt(cfg.d_model)))
    packed = nllb.pack_hf_state_dict(cfg, nllb.synth_state_dict(cfg, 0), 67)
    names, i = [], 0
    while True:
        name = C.c_char_p()
        if lib.wlk_nllb_tensor_name(C.byref(dims), i, C.byref(name)) != 0:
            break
        names.append(name.value.decode())
        i += 1
    assert sorted(names) == sorted(packed)
    for n in names:
        off, numel = C.c_uint64(), C.c_uint64()
        _lib.check(lib.wlk_nllb_tensor_lookup(C.byref(dims), n.encode(), C.byref(off), C.byref(numel)))
        assert numel.value == packed[n].size, n
    total = C.c_uint64()
    _lib.check(lib.wlk_nllb_arena_floats(C.byref(dims), C.byref(total)))
    assert total.value >= sum(v.size for v in packed.values())


# ---- the HIP library --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def micro_hip():
    m = nllb.HipNllbModel.synthetic(nllb.NLLB_MICRO, 0, device=0, max_src=92, max_tgt=64)
    yield m
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(N_CASES))
def test_hip_matches_transformers(micro_hip, ci):
    src, tgt = KAT[f"src{ci}"], KAT[f"tgt{ci}"]
    want = KAT[f"logits{ci}"]
    sess = micro_hip.new_session(1)
    try:
        sess.encode(src)
        np.testing.assert_allclose(sess.encoder_output(), KAT[f"enc{ci}"], rtol=0, atol=ENC_ATOL)
        sess.decode(tgt[None], first=True)                                    # the whole prefix at once
        np.testing.assert_allclose(sess.logits()[0], want[-1], rtol=0, atol=LOGIT_ATOL)
        for i, t in enumerate(tgt):                                           # token by token over the KV cache
            sess.decode(np.asarray([[int(t)]]), first=(i == 0))
            got = sess.logits()[0]
            np.testing.assert_allclose(got, want[i], rtol=0, atol=LOGIT_ATOL, err_msg=f"step {i}")
            lp, ids = sess.topk(4)
            ref = torch.log_softmax(torch.from_numpy(want[i]), dim=-1)
            np.testing.assert_allclose(lp[0], ref.topk(4)[0].numpy(), rtol=0, atol=LOGIT_ATOL)
            assert ids[0, 0] == int(ref.argmax()) or abs(float(ref[ids[0, 0]] - ref.max())) < H.TIE_EPS
    finally:
        sess.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(N_CASES))
def test_hip_generate_matches_transformers(micro_hip, ci):
    lang, max_new = (int(v) for v in KAT["cases"][ci])
    sess = micro_hip.new_session(1)
    try:
        assert nllb.generate(sess, KAT[f"src{ci}"], lang, max_new_tokens=max_new) == KAT[f"gen{ci}"].tolist()
        assert nllb.generate(sess, KAT[f"src{ci}"], lang, max_new_tokens=max_new,
                             forced_eos_token_id=nllb.NLLB_MICRO.eos_token_id) == KAT[f"gen_eos{ci}"].tolist()
    finally:
        sess.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bi", range(N_BEAM))
def test_hip_beam_search_matches_transformers(bi):
    lang, kw = _beam_args(bi)
    model = nllb.HipNllbModel.from_hf_state_dict(
        nllb.NLLB_MICRO, nllb.synth_state_dict(nllb.NLLB_MICRO, BEAM_WEIGHTS["seed"], BEAM_WEIGHTS["eos_gain"]), device=0,
        max_src=92, max_tgt=64)
    sess = model.new_session(kw["num_beams"])
    try:
        assert nllb.beam_search(sess, KAT[f"beam_src{bi}"], lang, **kw) == KAT[f"beam_out{bi}"].tolist()
    finally:
        sess.close()
        model.close()


@pytest.mark.gpu
def test_hip_rows_and_reorder(micro_hip):
    """Several hypothesis rows against one source: each row's logits equal the 1-row session's for the same prefix, and a
    cache reorder moves the prefixes with their rows."""
    src = KAT["src2"]
    prefixes = np.asarray([[2, 1992, 10, 11], [2, 1992, 500, 600], [2, 1992, 10, 600]], np.int64)
    one = micro_hip.new_session(1)
    many = micro_hip.new_session(3)
    try:
        one.encode(src); many.encode(src)
        many.decode(prefixes, first=True)
        got = many.logits()
        for r in range(3):
            one.decode(prefixes[r:r + 1], first=True)
            np.testing.assert_allclose(got[r], one.logits()[0], rtol=0, atol=1e-5)
        many.kv_reorder([2, 0, 0])
        nxt = np.asarray([[7], [8], [9]], np.int64)
        many.decode(nxt, first=False)
        got = many.logits()
        for r, srcrow in enumerate([2, 0, 0]):
            one.decode(np.concatenate([prefixes[srcrow], nxt[r]])[None], first=True)
            np.testing.assert_allclose(got[r], one.logits()[0], rtol=0, atol=2e-5)
        # the graph-replayed step (wlk_nllb_step) = decode of one token per row + top-k, after another reorder
        many.kv_reorder([1, 1, 2])
        lp, ids = many.step([20, 21, 22], 3)
        hist = [np.concatenate([prefixes[0], [8]]), np.concatenate([prefixes[0], [8]]), np.concatenate([prefixes[0], [9]])]
        for r, tok in enumerate([20, 21, 22]):
            one.decode(np.concatenate([hist[r], [tok]])[None], first=True)
            want_lp, want_ids = one.topk(3)
            np.testing.assert_allclose(lp[r], want_lp[0], rtol=0, atol=2e-5)
            assert ids[r].tolist() == want_ids[0].tolist()
        lp2, ids2 = many.step([30, 31, 32], 3)                                # replay of the captured graph
        for r, (a, b) in enumerate(zip([20, 21, 22], [30, 31, 32])):
            one.decode(np.concatenate([hist[r], [a, b]])[None], first=True)
            want_lp, want_ids = one.topk(3)
            np.testing.assert_allclose(lp2[r], want_lp[0], rtol=0, atol=2e-5)
            assert ids2[r].tolist() == want_ids[0].tolist()
    finally:
        one.close(); many.close()


@pytest.mark.gpu
def test_hip_session_reused_for_sources_of_different_lengths(micro_hip):
    """One session, several sentences (what a translation stream does): the graph-replayed step carries the source length of
    its recording - a sentence of another length must not replay it (round 5: it did, and the cross-attention read the
    previous sentence's key count).  Every generate on the reused session == the same generate on a fresh one."""
    reused = micro_hip.new_session(1)
    try:
        for ci in (1, 2, 0, 3, 1, 4, 2):                       # 5, 17, 1, 64, 5, 90, 17 source tokens
            lang, max_new = (int(v) for v in KAT["cases"][ci])
            fresh = micro_hip.new_session(1)
            try:
                want = nllb.generate(fresh, KAT[f"src{ci}"], lang, max_new_tokens=max_new)
            finally:
                fresh.close()
            assert want == KAT[f"gen{ci}"].tolist()
            assert nllb.generate(reused, KAT[f"src{ci}"], lang, max_new_tokens=max_new) == want, f"case {ci} on the reused session"
    finally:
        reused.close()


@pytest.mark.gpu
def test_hip_rejects_bad_input(micro_hip):
    from whisperlivekit_amd._lib import WlkError
    sess = micro_hip.new_session(1)
    try:
        with pytest.raises(WlkError):
            sess.decode(np.asarray([[2]]), first=True)                        # before an encode
        with pytest.raises(WlkError):
            sess.encode([5, 1, 6])                                            # padding inside the sentence
        with pytest.raises(WlkError):
            sess.encode([5] * 93)                                             # longer than max_src
        sess.encode([5, 6, 2])
        with pytest.raises(WlkError):
            sess.decode(np.asarray([[2]]), first=False)                       # a step before the prompt
        with pytest.raises(WlkError):
            sess.decode(np.asarray([[99999]]), first=True)
    finally:
        sess.close()


@pytest.mark.gpu
def test_hip_600m_shape_matches_the_oracle():
    """NLLB-200-distilled-600M's real dimensions (12 + 12 layers, 1024 wide, 16 heads, FFN 4096, 256 206 tokens) with
    seeded weights: encoder output and three decoder steps against the CPU oracle."""
    cfg = nllb.NLLB_200_DISTILLED_600M
    sd = nllb.synth_state_dict(cfg, 1)
    oracle = NllbOracle(cfg, sd)
    model = nllb.HipNllbModel.from_hf_state_dict(cfg, sd, device=0, max_src=64, max_tgt=32)
    sess = model.new_session(1)
    try:
        rng = np.random.default_rng(3)
        src = np.concatenate([[256047], rng.integers(4, 250000, size=21), [2]]).astype(np.int64)   # language tag, text, </s>
        sess.encode(src)
        enc = oracle.encode(src)
        np.testing.assert_allclose(sess.encoder_output(), enc.numpy(), rtol=0, atol=2e-3)
        cache = oracle.new_cache()
        prefix = np.asarray([[2, 256057]], np.int64)
        for i, toks in enumerate([prefix, np.asarray([[1234]]), np.asarray([[99]])]):
            sess.decode(toks, first=(i == 0))
            want = oracle.decode(torch.from_numpy(toks), enc, cache)[0, -1].numpy()
            got = sess.logits()[0]
            np.testing.assert_allclose(got, want, rtol=0, atol=5e-3, err_msg=f"step {i}")
            assert int(got.argmax()) == int(want.argmax()) or abs(want[int(got.argmax())] - want.max()) < 1e-3
    finally:
        sess.close()
        model.close()


# ---- the 600M shape against `transformers`' own M2M100 (scripts/gen_golden_nllb_600m.py -> tests/golden/nllb_600m_kat.npz)
def _kat_600m():
    z = np.load(os.path.join(H.GOLDEN, "nllb_600m_kat.npz"))
    steps, flat, off = [], z["step_tokens"], 0
    for n in z["steps"]:
        steps.append(flat[off:off + int(n)][None].astype(np.int64))
        off += int(n)
    return z, steps


def _check_600m_row(z, i, row, atol):
    probe = z["probe_ids"]
    np.testing.assert_allclose(row[probe], z[f"probe{i}"], rtol=0, atol=atol, err_msg=f"step {i}: probes")
    np.testing.assert_allclose(row[z[f"top_ids{i}"]], z[f"top_vals{i}"], rtol=0, atol=atol, err_msg=f"step {i}: top-16 values")
    m = float(row.max())
    lse = m + float(np.log(np.exp((row - m).astype(np.float64)).sum()))
    assert abs(lse - float(z[f"lse{i}"])) <= atol, (i, lse, float(z[f"lse{i}"]))
    best = int(row.argmax())
    assert best == int(z[f"top_ids{i}"][0]) or float(z[f"top_vals{i}"][0]) - float(row[int(z[f"top_ids{i}"][0])]) < atol


def test_oracle_600m_shape_matches_transformers():
    """oracle/nllb_oracle.py at NLLB-200-distilled-600M's dimensions == transformers' M2M100ForConditionalGeneration on the
    same seeded weights: encoder output and four cached decoder steps (top-16, log-sum-exp, 1 024 probe logits)."""
    z, steps = _kat_600m()
    cfg = nllb.NLLB_200_DISTILLED_600M
    torch.set_num_threads(8)
    oracle = NllbOracle(cfg, nllb.synth_state_dict(cfg, int(z["seed"])))
    enc = oracle.encode(z["src"])
    np.testing.assert_allclose(enc.numpy(), z["enc"], rtol=0, atol=2e-4)
    cache = oracle.new_cache()
    for i, toks in enumerate(steps):
        row = oracle.decode(torch.from_numpy(toks), enc, cache)[0, -1].numpy()
        _check_600m_row(z, i, row, 5e-4)


@pytest.mark.gpu
def test_hip_600m_shape_matches_transformers():
    """The HIP NLLB path at the 600M dimensions against the same transformers known answers, logits <= 1e-3."""
    z, steps = _kat_600m()
    cfg = nllb.NLLB_200_DISTILLED_600M
    model = nllb.HipNllbModel.from_hf_state_dict(cfg, nllb.synth_state_dict(cfg, int(z["seed"])), device=0, max_src=64, max_tgt=32)
    sess = model.new_session(1)
    try:
        sess.encode(z["src"])
        np.testing.assert_allclose(sess.encoder_output(), z["enc"], rtol=0, atol=1e-3)
        for i, toks in enumerate(steps):
            sess.decode(toks, first=(i == 0))
            _check_600m_row(z, i, sess.logits()[0], 1e-3)
    finally:
        sess.close()
        model.close()
