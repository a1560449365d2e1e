"""tools/campaign.py --reg-model physical-real-all without a GPU: a stand-in engine that only knows where a staging-register flip
lands (the matrix itself, the workgroup's next matrix, or -- an f piece requested in a matrix's last tile -- the one after that)
checks the campaign's bookkeeping: the staging flips run in a launch of their own, every wrong matrix is charged to the staging run
that caused it, the replica-private classes stay clean.  The real thing: tests/test_gpu_parity.py::
test_campaign_physical_real_all_registers_mm256."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Engine:
    """mm_batch returns zeros; a matrix a staging flip reaches gets one wrong word."""

    def __init__(self, ca):
        self.ca, self.armed, self.launches, self.seen = ca, None, 0, []

    def reset_stats(self):
        pass

    def inject_faults(self, f):
        self.armed = f

    def stats(self):
        return {"errors_corrected": 1, "sync_count": 1, "dwc_detected": 0}

    def last_launch(self):
        return {"engine": "matrix_core", "general_blocks": 0, "hooked_blocks": 0}

    def mm_batch(self, f, s, cfg=None, detected=None):
        self.launches += 1
        out = torch.zeros(f.shape[0], 4, dtype=torch.int32)
        if self.armed is not None and len(self.armed):
            self.seen.append(sorted({(int(r["step"]) >> 24) & 31 for r in self.armed if int(r["site"]) == self.ca.SITE_MM_VGPR}))
            for row in self.armed:
                m, reg, slab = int(row["item"]) // (256 * 256), (int(row["step"]) >> 24) & 31, int(row["step"]) & 3
                if int(row["site"]) != self.ca.SITE_MM_VGPR or reg < 12:
                    continue
                tgt = (m + 128 if slab == 3 else m + 64) if reg == 20 else (m + 64 if slab == 0 else m)
                if tgt < f.shape[0]:
                    out[tgt, 0] = 1
        self.armed = None
        return out


def _load(monkeypatch):
    def cpu(fn):
        def wrapped(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped

    for name in ("randint", "zeros", "Generator"):
        monkeypatch.setattr(torch, name, cpu(getattr(torch, name)))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    spec = importlib.util.spec_from_file_location("coast_campaign_cpu", os.path.join(ROOT, "tools", "campaign.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class MM(mod.MM):
        def inputs(self, runs, g):
            return [torch.zeros(runs, 2, 2, dtype=torch.int32), torch.zeros(runs, 2, 2, dtype=torch.int32)]

    monkeypatch.setitem(mod.BENCHES, "mm", MM)
    return mod


@pytest.mark.parametrize("model", ["physical-real", "physical-real-all"])
def test_physical_real_campaign_bookkeeping(monkeypatch, model):
    import coast_amd as ca

    mod = _load(monkeypatch)
    a = mod.parse(["-b", "mm", "--side", "256", "-m", "TMR", "-t", "600", "--reg-model", model, "-n"])
    eng = _Engine(ca)
    recs, summ = mod.run_campaign(a, eng)
    by = summ["by_class"]
    assert sum(d["runs"] for d in by.values()) == 600
    for cls in ("acc", "b_frag", "a_frag", "tally", "other"):
        assert by[cls]["errors"] == 0, by
    if model == "physical-real":  # the staging classes go through the model sites: one launch (after the golden run), no register >= 12
        assert eng.launches == 2 and all(max(regs, default=0) < 12 for regs in eng.seen)
        assert "errors_landed_in_a_later_matrix_of_the_workgroup" not in summ
        return
    assert eng.launches == 3
    assert max(eng.seen[0]) < 12 and min(eng.seen[1]) >= 12       # launch A: private classes; launch B: the staging flips alone
    assert summ["staging_launch_errors_without_a_staging_flip_to_blame"] == 0
    assert summ["errors_landed_in_a_later_matrix_of_the_workgroup"] > 0
    wrong = {r["run"] for r in recs if r["class"] == "error"}
    assert wrong and all(recs[r]["target"]["class"] in ("s_raw", "f_raw") for r in wrong)
    assert summ["errors"] == by["s_raw"]["errors"] + by["f_raw"]["errors"] == len(wrong)
    # every f_raw run whose landing matrix exists is an error; one without a flip (the dead share of the census) is not
    for r in recs:
        t = r["target"]
        if t["class"] == "f_raw":
            lands = r["run"] + (128 if (t["step"] & 3) == 3 else 64)
            assert (r["class"] == "error") == (lands < 600), (r["run"], t)


def test_uniform_campaign_isolates_the_runs_that_take_the_process_down(monkeypatch):
    """`--reg-model uniform` without a GPU: stand-in children that die (memory fault) or hang inside the launch that contains a `killer` run.
    The campaign must halve such a launch until the killer is alone, file exactly the killers as `invalid`, keep every other run's outcome,
    count the scalar class (s0..s101 and the spill registers' lanes) as errors without executing it, and cut a hanging child."""
    import json
    import subprocess

    mod = _load(monkeypatch)
    monkeypatch.setattr(mod, "kernel_registers", lambda replicas, clone=False, kernel="blocks3": (256, 102, [254, 255]))
    a = mod.parse(["-b", "mm", "--side", "256", "-m", "TMR", "-t", "700", "--reg-model", "uniform", "-n"])
    rng = np.random.default_rng(a.seed)
    draws = [mod.uniform_draw(rng, 256, 102) for _ in range(700)]
    scalar = {r for r, d in enumerate(draws) if d["file"] == 1 or d["reg"] in (254, 255)}
    vector = [r for r in range(700) if r not in scalar]
    crashers, hanger = {vector[5], vector[6], vector[300]}, vector[100]   # two in one launch, one elsewhere; and one that hangs
    wrong = {r for r in vector if r % 7 == 0} - crashers - {hanger}
    spawned = []

    class _Child:
        def __init__(self, argv, **kw):
            assert "--preg-child" in argv
            self.returncode = 0
            spawned.append(self)

        def communicate(self, spec=None, timeout=None):
            if spec is None:   # (after kill(): whatever was printed before)
                return self.out, None
            lines = []
            for k, grp in enumerate(json.loads(spec)["launches"]):
                runs = [r for r, _ in grp]
                lines.append("start %d" % k)
                if crashers & set(runs):
                    self.returncode = -6
                    break
                if hanger in runs:
                    self.out = "\n".join(lines) + "\n"
                    raise subprocess.TimeoutExpired("child", timeout)
                lines.append("done %d %s" % (k, json.dumps({"runs": [[r, int(r in wrong), int(r % 3 == 0)] for r in runs],
                                                           "stats": {"errors_corrected": len(runs), "sync_count": 1, "dwc_detected": 0}})))
            return "\n".join(lines) + "\n", None

        def kill(self):
            self.returncode = -9

    monkeypatch.setattr(subprocess, "Popen", _Child)
    recs, summ = mod.run_uniform_campaign(a)
    cls = {r["run"]: r["class"] for r in recs}
    assert {r for r, c in cls.items() if c == "invalid"} == crashers | {hanger}
    assert summ["invalids"] == 4 and summ["scalar_upsets_not_executed_counted_as_errors"] == len(scalar) > 0
    for r in range(700):
        if r in scalar:
            assert cls[r] == "error"
        elif r not in crashers and r != hanger:
            assert cls[r] == ("error" if r in wrong else "fault" if r % 3 == 0 else "success"), r
    assert summ["errors"] == len(scalar) + len(wrong) and summ["success"] + summ["faults"] + summ["errors"] + summ["invalids"] == 700
    assert abs(summ["coverage_pct"] - 100.0 * (700 - summ["errors"] - 4) / 700) < 1e-9
    assert 10 < len(spawned) < 40   # halving: about log2(64) children per killer, not one per run
