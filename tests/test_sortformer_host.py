"""a12 model side, CPU: the numpy speaker-cache / FIFO update of the product against the torch restatement in
oracle/sortformer_oracle.py, the weight re-packing algebra, the relative-position table and index map, and the
packed-layout contract of the C ABI (no GPU needed: layout queries do not touch HIP).

PARITY of the host part (speaker-cache update) is UNPINNED NeMo-side (see the oracle's header): these tests pin the two restatements against
each other and against closed-form properties, not against NeMo."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import sortformer_oracle as so
from whisperlivekit_amd import _lib
from whisperlivekit_amd import sortformer as sf

SMALL = sf.SortformerDims(n_mels=32, sub_channels=8, fc_d_model=32, fc_layers=2, fc_heads=2, fc_ff=64, conv_kernel=5,
                          tf_d_model=16, tf_layers=2, tf_heads=2, tf_inner=32, n_spk=4)


def as_torch(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def oracle_dims(d: sf.SortformerDims) -> so.SortformerDims:
    return so.SortformerDims(n_mels=d.n_mels, fc_d_model=d.fc_d_model, fc_layers=d.fc_layers, fc_heads=d.fc_heads,
                             conv_kernel=d.conv_kernel, sub_channels=d.sub_channels, tf_d_model=d.tf_d_model,
                             tf_layers=d.tf_layers, tf_heads=d.tf_heads, tf_inner=d.tf_inner, n_spk=d.n_spk)


def test_rel_shift_is_the_index_map_the_kernel_uses():
    """matrix_bd[i][j] after rel_shift == (q_i + v) . p[T-1-i+j]: sortformer.hip never materialises the shift."""
    torch.manual_seed(0)
    T, h, dk = 9, 2, 4
    qv, pp = torch.randn(h, T, dk), torch.randn(h, 2 * T - 1, dk)
    bd = so._rel_shift(torch.matmul(qv, pp.transpose(-2, -1)).unsqueeze(0))[0][:, :, :T]
    ref = torch.stack([torch.stack([(qv[:, i] * pp[:, T - 1 - i + j]).sum(-1) for j in range(T)], -1) for i in range(T)], 1)
    assert torch.allclose(bd, ref, atol=1e-6)


def test_positional_table_matches_oracle_and_is_centred():
    """A T-frame sequence uses rows [L-T, L+T-1) of the L-frame table: what wlk_sf_finalize relies on when it
    projects the table once for max_frames."""
    L, d = 40, 32
    big = sf.rel_positional_table(L, d)
    assert np.allclose(big, so.rel_positional_encoding(L, d).numpy(), atol=2e-6)
    for T in (1, 7, 40):
        assert np.array_equal(big[L - T: L + T - 1], sf.rel_positional_table(T, d))


def test_packed_layout_is_fully_provided_and_sized():
    cd = _lib.SfDims(SMALL.n_mels, SMALL.sub_channels, SMALL.fc_d_model, SMALL.fc_layers, SMALL.fc_heads, SMALL.fc_ff,
                     SMALL.conv_kernel, SMALL.tf_d_model, SMALL.tf_layers, SMALL.tf_heads, SMALL.tf_inner, SMALL.n_spk,
                     64, 128, 1.0)
    names = sf.packed_sortformer_names(cd)
    packed = sf.pack_sortformer_state_dict(SMALL, sf.synth_sortformer_state_dict(SMALL, 1), 64)
    assert set(names) == set(packed)
    lib = _lib.load()
    total = C.c_uint64()
    assert lib.wlk_sf_arena_floats(C.byref(cd), C.byref(total)) == 0
    end = 0
    for n in names:
        off, numel = C.c_uint64(), C.c_uint64()
        assert lib.wlk_sf_tensor_lookup(C.byref(cd), n.encode(), C.byref(off), C.byref(numel)) == 0
        assert numel.value == packed[n].size, n
        assert off.value % 64 == 0 and off.value >= end
        end = off.value + numel.value
    assert end <= total.value
    assert lib.wlk_sf_tensor_lookup(C.byref(cd), b"nope", None, None) != 0
    bad = _lib.SfDims(*([32, 8, 30, 2, 4] + [64, 5, 16, 2, 2, 32, 4, 64, 128]), 1.0)     # 30 % 4 heads
    assert lib.wlk_sf_arena_floats(C.byref(bad), C.byref(total)) != 0


def test_dims_round_trip_from_state_dict_shapes():
    d = sf.SortformerDims()
    shapes = {k: np.empty(v.shape, np.float32) for k, v in sf.synth_sortformer_state_dict(
        sf.SortformerDims(fc_layers=1, tf_layers=1), 0).items()}
    got = sf.dims_from_state_dict(shapes)
    assert (got.n_mels, got.sub_channels, got.fc_d_model, got.fc_heads, got.fc_ff, got.conv_kernel, got.tf_d_model,
            got.tf_inner, got.n_spk) == (d.n_mels, d.sub_channels, d.fc_d_model, d.fc_heads, d.fc_ff, d.conv_kernel,
                                         d.tf_d_model, d.tf_inner, d.n_spk)


def packed_forward_numpy(dims, pk, feats, L):
    """The launch sequence of sortformer_api.hip in numpy (fp64) on the PACKED tensors: proves the re-packings
    (tap-major convs, (freq, channel) feature order, folded 0.5, concatenated qkv, centred position rows)."""
    f8 = lambda a: np.asarray(a, np.float64)
    sub = lambda n: (n - 1) // 2 + 1
    T, F, Cc, d = feats.shape[0], dims.n_mels, dims.sub_channels, dims.fc_d_model

    def conv_s2(x, w_tap, b, depthwise):          # x [Ti, Fi, Cin or 1]; w_tap [9, C]
        Ti, Fi = x.shape[:2]
        xp = np.zeros((Ti + 2, Fi + 2, x.shape[2]))
        xp[1:-1, 1:-1] = x
        out = np.zeros((sub(Ti), sub(Fi), Cc))
        for ky in range(3):
            for kx in range(3):
                out += xp[ky: ky + 2 * sub(Ti): 2, kx: kx + 2 * sub(Fi): 2][:, :, : x.shape[2]] * w_tap[ky * 3 + kx]
        return out + b

    x = np.maximum(conv_s2(f8(feats)[:, :, None], f8(pk["pre.conv0.w"]).T, f8(pk["pre.conv0.b"]), False), 0)
    for i in (1, 2):
        x = conv_s2(x, f8(pk[f"pre.dw{i}.w"]), f8(pk[f"pre.dw{i}.b"]), True)
        x = np.maximum(x @ f8(pk[f"pre.pw{i}.w"]).T + f8(pk[f"pre.pw{i}.b"]), 0)
    emb = x.reshape(x.shape[0], -1) @ f8(pk["pre.out.w"]).T + f8(pk["pre.out.b"])
    T = emb.shape[0]
    ln = lambda v, w, b: (v - v.mean(-1, keepdims=True)) / np.sqrt(v.var(-1, keepdims=True) + 1e-5) * f8(w) + f8(b)
    sig = lambda v: 1 / (1 + np.exp(-v))
    x = emb * (math.sqrt(d) if dims.xscaling else 1.0)
    for l in range(dims.fc_layers):
        g = lambda n: f8(pk[f"fc.{l}.{n}"])
        for ff in ("ff1",):
            h = ln(x, g("ln_ff1.w"), g("ln_ff1.b")) @ g("ff1a.w").T + g("ff1a.b")
            x = x + (h * sig(h)) @ g("ff1b.w").T + g("ff1b.b")
        y = ln(x, g("ln_att.w"), g("ln_att.b")) @ g("qkv.w").T + g("qkv.b")
        H, dk = dims.fc_heads, d // dims.fc_heads
        q, k, v = y[:, :d], y[:, d:2 * d], y[:, 2 * d:]
        P = f8(pk["pos.table"]) @ g("pos.w").T
        ctx = np.zeros((T, d))
        for hh in range(H):
            s_ = slice(hh * dk, (hh + 1) * dk)
            sc = np.zeros((T, T))
            for i in range(T):
                for j in range(T):
                    sc[i, j] = ((q[i, s_] + g("bias_u")[s_]) @ k[j, s_] + (q[i, s_] + g("bias_v")[s_]) @ P[L - 1 - i + j, s_]) / math.sqrt(dk)
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            ctx[:, s_] = (sc / sc.sum(-1, keepdims=True)) @ v[:, s_]
        x = x + ctx @ g("out.w").T + g("out.b")
        y = ln(x, g("ln_conv.w"), g("ln_conv.b")) @ g("pw1.w").T + g("pw1.b")
        glu = y[:, :d] * sig(y[:, d:])
        half = (dims.conv_kernel - 1) // 2
        gp = np.zeros((T + 2 * half, d))
        gp[half: half + T] = glu
        acc = sum(gp[kk: kk + T] * g("dw.w")[kk] for kk in range(dims.conv_kernel)) + g("dw.b")
        bn = (acc - g("bn.mean")) * g("bn.invstd") * g("bn.w") + g("bn.b")
        x = x + (bn * sig(bn)) @ g("pw2.w").T + g("pw2.b")
        h = ln(x, g("ln_ff2.w"), g("ln_ff2.b")) @ g("ff2a.w").T + g("ff2a.b")
        x = x + (h * sig(h)) @ g("ff2b.w").T + g("ff2b.b")
        x = ln(x, g("ln_out.w"), g("ln_out.b"))
    dt, Ht = dims.tf_d_model, dims.tf_heads
    x = x @ f8(pk["proj.w"]).T + f8(pk["proj.b"])
    for l in range(dims.tf_layers):
        g = lambda n: f8(pk[f"tf.{l}.{n}"])
        y = x @ g("qkv.w").T + g("qkv.b")
        dh = dt // Ht
        y[:, : 2 * dt] *= dh ** -0.25
        ctx = np.zeros((T, dt))
        for hh in range(Ht):
            s_ = slice(hh * dh, (hh + 1) * dh)
            sc = y[:, :dt][:, s_] @ y[:, dt:2 * dt][:, s_].T
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            ctx[:, s_] = (sc / sc.sum(-1, keepdims=True)) @ y[:, 2 * dt:][:, s_]
        x = ln(ctx @ g("out.w").T + g("out.b") + x, g("ln1.w"), g("ln1.b"))
        x = ln(np.maximum(x @ g("in.w").T + g("in.b"), 0) @ g("outd.w").T + g("outd.b") + x, g("ln2.w"), g("ln2.b"))
    h = np.maximum(np.maximum(x, 0) @ f8(pk["head.h.w"]).T + f8(pk["head.h.b"]), 0)
    return emb, sig(h @ f8(pk["head.s.w"]).T + f8(pk["head.s.b"]))


def test_packing_preserves_the_network_function():
    sd = sf.synth_sortformer_state_dict(SMALL, 3)
    L = 24
    pk = sf.pack_sortformer_state_dict(SMALL, sd, L)
    feats = np.random.default_rng(0).standard_normal((61, SMALL.n_mels)).astype(np.float32)
    emb, preds = packed_forward_numpy(SMALL, pk, feats, L)
    od, tsd = oracle_dims(SMALL), as_torch(sd)
    emb_o = so.pre_encode(tsd, od, torch.from_numpy(feats))
    assert emb.shape == tuple(emb_o.shape) == (8, SMALL.fc_d_model)
    assert np.allclose(emb, emb_o.numpy(), atol=2e-5)
    preds_o = so.forward_embeddings(tsd, od, emb_o).numpy()
    assert np.allclose(preds, preds_o, atol=2e-5)
    assert preds_o.std() > 0.01          # synthetic weights give non-degenerate activities


def random_stream(seed, n_steps, d=16, n_spk=4):
    rng = np.random.default_rng(seed)
    for step in range(n_steps):
        tc = 13 if step == 0 else 25
        yield step, rng.standard_normal((tc, d)).astype(np.float32), rng


@pytest.mark.parametrize("seed,silence", [(0, False), (1, True), (2, False)])
def test_numpy_streaming_update_equals_the_oracle(seed, silence):
    """40 steps: FIFO overflow, cache growth, repeated compression, silence profile."""
    d, n_spk = 16, 4
    sp_n, sp_t = sf.SpkCacheParams(), so.StreamParams()
    st_n = sf.SortformerState(np.zeros((188, d), np.float32), np.zeros((188, n_spk), np.float32),
                              np.zeros((188, d), np.float32), np.zeros((188, n_spk), np.float32), np.zeros(d, np.float32))
    st_t = so.new_stream_state(so.SortformerDims(fc_d_model=d, n_spk=n_spk), sp_t)
    compressed = ties = 0
    for step, chunk, rng in random_stream(seed, 40, d, n_spk):
        T = st_n.spkcache_len + st_n.fifo_len + chunk.shape[0]
        preds = rng.random((T, n_spk)).astype(np.float32) ** (3.0 if silence else 1.0)
        if silence:
            preds[rng.random(T) < 0.3] *= 0.02
        lc, rc = (0 if step == 0 else 1), 1
        before = st_n.spkcache_len
        out_n = sf.streaming_update(sp_n, st_n, chunk, preds, lc, rc)
        out_t = so.streaming_update(sp_t, st_t, torch.from_numpy(chunk), torch.from_numpy(preds), lc, rc)
        compressed += int(before + 144 > 188 and st_n.spkcache_len == 188 and before != st_n.spkcache_len or
                          (before == 188 and st_n.fifo_len < 100))
        assert np.array_equal(out_n, out_t.numpy())
        assert (st_n.spkcache_len, st_n.fifo_len, st_n.n_sil_frames) == (st_t.spkcache_len, st_t.fifo_len, st_t.n_sil_frames)
        if not (np.allclose(st_n.spkcache, st_t.spkcache.numpy(), atol=1e-6)
                and np.allclose(st_n.spkcache_preds, st_t.spkcache_preds.numpy(), atol=1e-6)):
            # The cache holds duplicates (one frame kept under several speakers), so the global top-k can cut
            # through EXACTLY tied scores; torch.topk's choice among ties is unspecified (the product takes the
            # lowest index).  Then the two caches must still be equal as multisets of (embedding, activity) rows.
            rows_n = np.concatenate([st_n.spkcache, st_n.spkcache_preds], 1)
            rows_t = np.concatenate([st_t.spkcache.numpy(), st_t.spkcache_preds.numpy()], 1)
            key = lambda r: r[np.lexsort(np.round(r, 4).T[::-1])]
            assert np.allclose(key(rows_n), key(rows_t), atol=1e-5), step
            ties += 1
            st_t.spkcache, st_t.spkcache_preds = torch.from_numpy(st_n.spkcache.copy()), torch.from_numpy(st_n.spkcache_preds.copy())
        assert np.allclose(st_n.fifo, st_t.fifo.numpy(), atol=1e-6)
        assert np.allclose(st_n.fifo_preds, st_t.fifo_preds.numpy(), atol=1e-6)
        assert np.allclose(st_n.mean_sil_emb, st_t.mean_sil_emb.numpy(), atol=1e-6)
    assert st_n.spkcache_len == 188 and compressed >= 2 and ties <= 3
    if silence:
        assert st_n.n_sil_frames > 0


def test_compression_keeps_every_speaker_and_time_order():
    """Closed-form properties of _compress_spkcache: the output has spkcache_len rows; rows are grouped by
    speaker and time-ordered within a speaker; silence slots carry the mean silence embedding and zero activity."""
    rng = np.random.default_rng(5)
    sp = sf.SpkCacheParams()
    n, d = 188 + 144, 8
    emb = np.arange(n, dtype=np.float32)[:, None].repeat(d, 1)           # embedding value == frame index
    preds = np.zeros((n, 4), np.float32)
    for s in range(4):
        preds[s * 80: s * 80 + 70, s] = 0.6 + 0.39 * rng.random(70)
    sil = np.full(d, -1.0, np.float32)
    e, p = sf.compress_spkcache(sp, emb, preds, sil)
    assert e.shape == (188, d) and p.shape == (188, 4)
    is_sil = e[:, 0] == -1.0
    assert np.all(p[is_sil] == 0) and is_sil.sum() >= 12
    owner = p[~is_sil].argmax(1)
    assert set(owner) == {0, 1, 2, 3}
    assert np.all(np.diff(owner) >= 0)
    for s in range(4):
        assert np.all(np.diff(e[~is_sil][owner == s, 0]) > 0)


def test_product_module_does_not_import_the_oracle():
    import whisperlivekit_amd.sortformer as mod
    src = open(mod.__file__).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_load_nemo_checkpoint_from_a_nemo_archive(tmp_path):
    """``.nemo`` = tar with model_weights.ckpt (a torch state dict) + model_config.yaml: the loader must find the
    weights whatever prefix the archive uses, recover the geometry from the tensor shapes, and the packer must place
    every tensor (what HipSortformerModel.from_checkpoint does before it touches the GPU)."""
    import io
    import tarfile

    import torch

    from whisperlivekit_amd import sortformer as sf
    dims = sf.SortformerDims(fc_layers=2, tf_layers=3)
    sd = sf.synth_sortformer_state_dict(dims, 5)
    blob = io.BytesIO()
    state = {k: torch.from_numpy(np.asarray(v)).half() if i % 7 == 0 else torch.from_numpy(np.asarray(v))
             for i, (k, v) in enumerate(sd.items())}          # a few fp16 tensors: the loader widens to fp32
    state["some.int.buffer"] = torch.arange(4)                # non-float entries are skipped
    torch.save(state, blob)
    path = tmp_path / "diar_test.nemo"
    with tarfile.open(path, "w") as tar:
        for name, data in (("./model_config.yaml", b"sample_rate: 16000\n"), ("./abc123_model_weights.ckpt", blob.getvalue())):
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
    got = sf.load_nemo_checkpoint(str(path))
    assert set(got) == set(sd) and all(v.dtype == np.float32 for v in got.values())
    for k in sd:
        tol = 2e-3 if state[k].dtype == torch.float16 else 0.0
        assert np.abs(got[k] - sd[k]).max() <= tol * max(1.0, float(np.abs(sd[k]).max())), k
    d2 = sf.dims_from_state_dict(got)
    assert (d2.fc_layers, d2.tf_layers, d2.fc_d_model, d2.tf_d_model, d2.n_spk, d2.n_mels) == \
           (2, 3, dims.fc_d_model, dims.tf_d_model, dims.n_spk, dims.n_mels)
    packed = sf.pack_sortformer_state_dict(d2, got, max_frames=64)
    assert all(np.isfinite(v).all() for v in packed.values()) and "pre.out.w" in packed
