import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def ok(lo, hi):
    env = dict(os.environ, COAST_CAMPAIGN_ONLY="%d:%d" % (lo, hi))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "campaign.py"), "-b", "mm", "--side", "256", "-m", "TMR", "-t", "5000",
                        "--reg-model", "uniform", "-n"], env=env, capture_output=True, text=True, timeout=200)
    return p.returncode == 0 and "Coverage" in p.stdout
lo, hi = 0, 5000
found = []
for attempt in range(3):
    lo, hi = (found[-1] + 1 if found else 0), 5000
    if ok(lo, hi):
        break
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if ok(lo, mid):
            lo = mid
        else:
            hi = mid
    found.append(lo)
    print("crashing run", lo, flush=True)
# print the draws of the crashing runs
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, importlib.util
spec = importlib.util.spec_from_file_location("camp", os.path.join(ROOT, "tools", "campaign.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
rng = np.random.default_rng(0)
draws = [m.uniform_draw(rng) for _ in range(5000)]
for r in found:
    print(r, json.dumps(draws[r]))
