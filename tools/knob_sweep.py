"""Bench lines under environment variations: python tools/knob_sweep.py "NAME=VAL,NAME2=VAL2" "..." -- prints value, contour/rr ms, factor_ms, store=false rate.
Every variation runs bench.py in a child process (the switches are read at handle creation)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = os.environ.get("SWEEP_ARGS", "--cpu-baseline off --cold-solves 0 --store-false-steps 0 --steps 10 --warmup 3").split()
for spec in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=", 1); env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{spec or 'default':50s} it/s {d['value']:.2f} step {d['ms_per_step']:.2f} ms apply {d['roofline']['kernel_ms_per_step']:.2f} ms launches {d['roofline']['launches_per_step']} "
              f"factor {d['setup']['factor_ms']:.2f} ms store=false {d['setup']['store_false_iterations_per_s']:.2f} analyze {d['setup']['analyze_s']*1e3:.0f} ms", flush=True)
    except Exception as e:
        print(spec, "FAILED", e, out.stderr[-400:], flush=True)
