"""The reference's package from the staged archive (oracle/_ref/, oracle/stage_reference.py): what the GPU box has instead of
/root/reference.  `scripts/ref_stubs.py` must find it, unpack it outside the repository and import the reference's
simul_whisper classes from there - the route the GPU drop-in tests and bench.py's cpu_baseline leg take on that box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stage_reference  # noqa: E402

pytestmark = pytest.mark.skipif(not stage_reference.staged(), reason="no staged archive (oracle/stage_reference.py has not run)")


def test_ref_stubs_falls_back_to_the_staged_archive():
    code = r"""
import os, os.path as op, sys
real = op.isdir
op.isdir = lambda p: False if str(p).startswith('/root/reference') else real(p)     # the GPU box has no such tree
sys.path.insert(0, os.path.join(%r, 'scripts'))
import ref_stubs
assert ref_stubs.reference_available(), ref_stubs.REFERENCE_ROOT
root = ref_stubs.REFERENCE_ROOT
assert not root.startswith(%r) and not root.startswith('/root/reference'), root    # unpacked OUTSIDE the repository
ref_stubs.install()
import whisperlivekit
from whisperlivekit.simul_whisper.backend import SimulStreamingOnlineProcessor
from whisperlivekit.simul_whisper.align_att_base import AlignAttBase
assert whisperlivekit.__file__.startswith(root)
print('ok', root)
""" % (ROOT, ROOT)
    env = {k: v for k, v in os.environ.items() if k != "WLK_REFERENCE_ROOT"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1].startswith("ok ")


def test_archive_manifest_matches_its_contents():
    import json
    import tarfile
    man = json.load(open(stage_reference.MANIFEST))
    with tarfile.open(stage_reference.ARCHIVE, "r:gz") as tar:
        names = [m.name for m in tar.getmembers()]
    assert len(names) == man["files"]
    assert all(n.startswith("whisperlivekit/") for n in names)
    assert not any(n.endswith((".pt", ".jit", ".onnx", ".bin")) for n in names)     # sources and small data files only
