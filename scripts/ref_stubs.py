"""Harness-side stand-ins that let the *unmodified* reference package import in the build
container.  Nothing here is product code: the three modules below are absent
from this image, none of them touches the numerics of the simul_whisper path:

* ``soundfile`` - only used by file-based ASR wrappers (local_agreement/backends.py:8);
* ``numba``     - only decorates the offline DTW (whisper/timing.py:57,82);
* ``tiktoken``  - backs ``Tokenizer``; replaced by this repo's encodings so that the reference
  and the HIP backend split ids into words with the same vocabulary.

Used by ``scripts/gen_golden.py`` and the ``tests/test_reference_*`` / ``test_gpu_reference_dropin`` tests; on the GPU box
(no /root/reference) the reference comes from the archive staged under ``oracle/_ref/`` - the tests skip only when that
is missing too.
"""
import os
import sys
import types

def _resolve_reference_root() -> str:
    """WLK_REFERENCE_ROOT, else /root/reference (the build container), else the archive of the reference's own sources that
    `oracle/stage_reference.py` stages under git-ignored `oracle/_ref/` (it travels to the GPU box like a built `.so`):
    unpacked once per process into a temporary directory outside the repository."""
    env = os.environ.get("WLK_REFERENCE_ROOT")
    if env and os.path.isdir(os.path.join(env, "whisperlivekit")):
        return env
    if os.path.isdir("/root/reference/whisperlivekit"):
        return "/root/reference"
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    try:
        from oracle import stage_reference
        if stage_reference.staged():
            import atexit
            import shutil
            import tempfile
            dst = tempfile.mkdtemp(prefix="wlk_ref_")
            atexit.register(shutil.rmtree, dst, ignore_errors=True)
            return stage_reference.unpack(dst)
    except Exception:        # a broken archive = no reference: the tests that need it skip
        pass
    return env or "/root/reference"


REFERENCE_ROOT = _resolve_reference_root()


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "whisperlivekit"))


def install(synthetic_vocab: bool = True):
    """Insert the stub modules and put the reference on sys.path.  Idempotent."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from whisperlivekit_amd import tokenizer as wtok

    if "soundfile" not in sys.modules:
        sf = types.ModuleType("soundfile")
        def _no(*a, **k):
            raise RuntimeError("soundfile stub")
        sf.read = sf.write = _no
        sys.modules["soundfile"] = sf

    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        def jit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        nb.jit = jit
        sys.modules["numba"] = nb

    if "tiktoken" not in sys.modules:
        tk = types.ModuleType("tiktoken")

        class Encoding:  # tiktoken.Encoding(name=, explicit_n_vocab=, pat_str=, mergeable_ranks=, special_tokens=)
            def __init__(self, name, explicit_n_vocab=None, pat_str=None, mergeable_ranks=None,
                         special_tokens=None):
                specials = [k for k, _ in sorted(special_tokens.items(), key=lambda kv: kv[1])]
                if synthetic_vocab:
                    self._e = wtok.SyntheticEncoding(len(mergeable_ranks), specials, name=name)
                else:
                    self._e = wtok.BpeEncoding(dict(mergeable_ranks), specials, name=name)
                assert self._e.special_tokens == dict(special_tokens)
                self.name = name
                self.n_vocab = self._e.n_vocab

            @property
            def special_tokens_set(self):
                return self._e.special_tokens_set

            @property
            def eot_token(self):
                return self._e.eot_token

            def encode_single_token(self, t):
                return self._e.encode_single_token(t)

            def encode(self, text, **kw):
                return self._e.encode(text)

            def decode(self, ids, **kw):
                return self._e.decode(ids)

        tk.Encoding = Encoding
        sys.modules["tiktoken"] = tk

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
