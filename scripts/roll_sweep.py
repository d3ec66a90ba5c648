#!/usr/bin/env python3
"""Sweep the persistent rollout's tile size / grid (TRON_ROLL_E, TRON_ROLL_GRID) through bench.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
combos = [tuple(int(x) for x in c.split("x")) for c in sys.argv[1:]] or [(32, 0)]
for combo in combos:
    e, g = combo[:2]
    env = dict(os.environ, TRON_ROLL_E=str(e))
    if g:
        env["TRON_ROLL_GRID"] = str(g)
    if len(combo) > 2:
        env["TRON_ROLL_CHUNK"] = str(combo[2])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "640", "--warmup", "64"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    print(f"E={e:3d} grid={g or 'auto':>5} chunk={combo[2] if len(combo) > 2 else 64:4d}  {d['ms_per_step'] * 1e3:7.2f} us/step  {d['value'] / 1e9:6.3f} G env-steps/s  frac {d['roofline']['frac']:.3f}",
          flush=True)
