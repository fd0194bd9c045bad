"""The drop-in proven on the GPU: the REFERENCE's own SimulStreamingOnlineProcessor (only _create_alignatt
overridden - whisperlivekit/simul_whisper/backend.py:61-71, the routing hook INTEGRATION.md describes) with the
REFERENCE's own AlignAttBase.infer as the policy, over the real HIP hooks and a real HipWhisperModel, replaying golden
streams the unmodified reference produced.  The reference's package comes from WLK_REFERENCE_ROOT, /root/reference, or -
on the GPU box - from the archive `oracle/stage_reference.py` stages under git-ignored `oracle/_ref/` (test
infrastructure, like the CPU baseline of bench.py; scripts/ref_stubs.py unpacks it outside the repository)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_stubs  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_stubs.reference_available(), reason="no WhisperLiveKit tree (WLK_REFERENCE_ROOT, /root/reference or oracle/_ref)")]

import helpers as H  # noqa: E402

_models = {}


def _hip_model(name, seed):
    from whisperlivekit_amd.engine import HipWhisperModel
    if (name, seed) not in _models:
        _models[(name, seed)] = HipWhisperModel.synthetic(name, seed)
    return _models[(name, seed)]


@pytest.mark.parametrize("case", ["micro_12s", "micro_34s_evict", "micro_beam2", "micro_events", "micro_minlen_beam3", "micro_single_35s",
                                  "micromulti_auto", "tiny_6s", "base_4s", "bench_base_30s_s0"])
def test_reference_processor_and_policy_over_real_hip_hooks(case):
    from test_oracle_golden import check_stream_against_golden, replay_stream
    from test_reference_dropin import _make
    if not H.golden_exists(f"stream_{case}.json"):
        pytest.skip(f"golden stream {case} not generated")
    ref_stubs.install(synthetic_vocab=True)       # puts the reference package on sys.path (idempotent; _make does it too, but later)
    from whisperlivekit.simul_whisper.align_att_base import AlignAttBase
    from whisperlivekit.simul_whisper.backend import SimulStreamingOnlineProcessor
    from whisperlivekit_amd.engine import HipSession
    opened = []

    def run(teacher):
        def mk(m, c, seed=0):
            proc = _make(m, c, seed, model_factory=_hip_model)
            proc.model.teacher = dict(teacher) or None       # fp32 ties: the reference's side is forced on a replay
            return proc
        g, proc, got = replay_stream(case, mk)
        opened.append(proc)
        assert isinstance(proc, SimulStreamingOnlineProcessor)            # the reference's session object
        assert type(proc).process_iter is SimulStreamingOnlineProcessor.process_iter
        assert isinstance(proc.model, AlignAttBase) and type(proc.model).infer is AlignAttBase.infer
        assert isinstance(proc.model.session, HipSession)                  # real C-ABI session, not the CPU fake
        return g, proc, got

    (g, proc, got), ties = H.run_resynced(run, H.golden_json(f"stream_{case}.json"),
                                          lambda res: check_stream_against_golden(res[0], res[1].trace, res[2], tol=1e-3,
                                                                                  allow_ties=True))
    n = sum(len(r["steps"]) for r in proc.trace)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "dropin_gpu_report.txt"), "a") as fh:
        fh.write(f"{case}: reference SimulStreamingOnlineProcessor + AlignAttBase.infer over HIP hooks, {n} decode steps, "
                 f"{len(g['calls'])} calls all compared, ties re-synchronised: {ties}\n")
    for p in opened[:-1]:
        p.model.close()
    proc.model.close()
