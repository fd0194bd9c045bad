#!/usr/bin/env python
"""Recipe for `oracle/_ref/`: the reference's own Python package, packed where it lies.

    python oracle/stage_reference.py [--root /root/reference]

TEST INFRASTRUCTURE ONLY.  The reference (WhisperLiveKit, a pure-Python package) cannot be compiled into a binary the
way a C reference would be; what travels to the GPU box instead is ONE archive of its unmodified sources,
`oracle/_ref/wlk_reference.tar.gz` (git-ignored like every built artefact - never part of the history, never
unpacked inside the repository).  `bench.py`'s `cpu_baseline` leg unpacks it into a temporary directory, imports the
reference's `SimulStreamingOnlineProcessor` from there with the three harness stubs of `scripts/ref_stubs.py`, runs it
on the host CPU and validates what it emits against the golden stream the same code produced in the build container
(`cpu_baseline.kind = "reference"`, `validated: true`).  Nothing under `whisperlivekit_amd/` ever looks at it.

`__graft_entry__.build()` calls `stage()` whenever /root/reference is present (the build container); on the GPU box the
prebuilt archive is used as it arrived.

What goes in: every `*.py` of the `whisperlivekit` package plus the small data files the simul_whisper path opens at
import / run time (`whisper/assets/*`: mel filterbank, rank tables; `simul_whisper/` support files).  What stays out:
model binaries (`silero_vad_models/*`, 6.9 MB), the web front end, byte-code caches.
"""
import gzip
import hashlib
import io
import json
import os
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "wlk_reference.tar.gz")
MANIFEST = os.path.join(OUT_DIR, "MANIFEST.json")

SKIP_DIRS = {"__pycache__", "silero_vad_models"}
KEEP_EXT = {".py", ".npz", ".tiktoken", ".json", ".txt", ".html", ".css", ".js", ".svg"}


def _wanted(rel: str) -> bool:
    parts = rel.split(os.sep)
    if any(p in SKIP_DIRS for p in parts):
        return False
    return os.path.splitext(rel)[1] in KEEP_EXT


def stage(root: str = "/root/reference", verbose: bool = True) -> str:
    pkg = os.path.join(root, "whisperlivekit")
    if not os.path.isdir(pkg):
        raise FileNotFoundError(f"no WhisperLiveKit tree at {root}")
    os.makedirs(OUT_DIR, exist_ok=True)
    files = []
    for base, dirs, names in os.walk(pkg):
        dirs[:] = sorted(d for d in dirs if d not in SKIP_DIRS)
        for n in sorted(names):
            full = os.path.join(base, n)
            rel = os.path.relpath(full, root)
            if _wanted(rel):
                files.append((rel, full))
    digest = hashlib.sha256()
    buf = io.BytesIO()
    # the gzip header carries a timestamp of its own: pinned to 0 like the members' (tarfile's "w:gz" would stamp now)
    with gzip.GzipFile(filename="", fileobj=buf, mode="wb", compresslevel=6, mtime=0) as gz, \
            tarfile.open(fileobj=gz, mode="w") as tar:
        for rel, full in files:
            with open(full, "rb") as fh:
                data = fh.read()
            digest.update(rel.encode() + b"\0" + data)
            info = tarfile.TarInfo(rel)
            info.size = len(data)
            info.mtime = 0            # reproducible archive: same tree -> same bytes
            info.mode = 0o644
            tar.addfile(info, io.BytesIO(data))
    version = None
    try:
        import re
        m = re.search(r'^version\s*=\s*"([^"]+)"', open(os.path.join(root, "pyproject.toml")).read(), re.M)
        version = m.group(1) if m else None
    except OSError:
        pass
    with open(ARCHIVE, "wb") as fh:
        fh.write(buf.getvalue())
    json.dump(dict(source=root, version=version, files=len(files), sha256_of_contents=digest.hexdigest(),
                   archive_bytes=len(buf.getvalue())), open(MANIFEST, "w"), indent=1)
    if verbose:
        print(f"staged {len(files)} files of WhisperLiveKit {version} -> {ARCHIVE} ({len(buf.getvalue()) / 1e6:.1f} MB)",
              file=sys.stderr)
    return ARCHIVE


def staged() -> bool:
    return os.path.isfile(ARCHIVE)


def unpack(dst: str) -> str:
    """Unpack the staged archive under ``dst`` (a temporary directory OUTSIDE the repository) -> reference root."""
    with tarfile.open(ARCHIVE, "r:gz") as tar:
        for m in tar.getmembers():
            if m.name.startswith("/") or ".." in m.name.split("/"):
                raise RuntimeError("unsafe path in archive: " + m.name)
        tar.extractall(dst)
    return dst


if __name__ == "__main__":
    root = sys.argv[sys.argv.index("--root") + 1] if "--root" in sys.argv else "/root/reference"
    print(stage(root))
