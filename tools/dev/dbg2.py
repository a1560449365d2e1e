import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = {"file": 0, "reg": 127, "lane": 11, "bit": 18, "wave": 3, "panel": 0, "step": 9, "slot": 41}
vars_ = [{}, {"bit": 3}, {"bit": 31}, {"lane": 12}, {"slot": 40}, {"slot": 42}, {"reg": 126}, {"reg": 128}, {"reg": 30}, {"wave": 2}, {"step": 8}, {"panel": 1}]
for v in vars_:
    d = dict(base, **v)
    spec = json.dumps({"mode": "TMR", "seed": 0, "launches": [[[0, d]]]})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "campaign.py"), "--preg-child", "-"], input=spec, capture_output=True, text=True, timeout=120)
    last = [l for l in p.stdout.splitlines() if l.startswith("done")]
    print(v, "rc", p.returncode, last[-1][:80] if last else "DIED", flush=True)
