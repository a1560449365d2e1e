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
