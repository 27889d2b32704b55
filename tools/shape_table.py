#!/usr/bin/env python3
"""Per-shape kernel table of one eager MCGAN iteration (event-timed, ops.profile_step with MCGEN_PROF_SHAPES=1):
which launches the step's time goes to.  usage (GPU box): MCGEN_PROF_SHAPES=1 python tools/shape_table.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, MCGEN_PROF_SHAPES='1')
out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--sustain-steps', '0',
                      '--steps', '5', '--warmup', '2'] + sys.argv[1:], env=env, capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
r = json.loads(line)['roofline']
rows = sorted(r['by_kernel'].items(), key=lambda kv: -kv[1]['total_ms'])
tot = sum(v['total_ms'] for _, v in rows)
print(f'MFMA kernels: {tot:.3f} ms over the profiled iterations')
for k, v in rows:
    print(f"{k:75s} x{v['launches']:5.1f} {v['total_ms']:7.3f} ms  {v['total_ms'] / v['launches'] * 1e3:7.1f} us  {v['tflops']:7.1f} TF  {v['gbytes_per_s']:7.1f} GB/s")
