#!/usr/bin/env python3
"""Run bench.py over a grid of (env, args) settings and print one compact line each (GPU box helper)."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def run(env, args):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args,
                         env=e, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.split("\n") if l.startswith("{")]
    if not line:
        print(env, args, "FAILED", out.stderr[-400:]); return
    d = json.loads(line[-1])
    ph = {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["phases"].items()}
    print(env, " ".join(args), "| %.1f pairs/s %.2f ms/step | frac %.3f |" % (d["value"], d["ms_per_step"], d["roofline"]["whole_path"]["frac"]), ph, flush=True)

if __name__ == "__main__":
    spec = json.loads(sys.argv[1])   # {"env": {"K": [..]}, "args": [[...], ...]}
    keys = list(spec.get("env", {}))
    for combo in itertools.product(*[spec["env"][k] for k in keys]):
        for a in spec.get("args", [[]]):
            run({k: str(v) for k, v in zip(keys, combo)}, [str(x) for x in a])
