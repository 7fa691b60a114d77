#!/usr/bin/env python3
"""Interleaved A/B of library builds / environment knobs through bench.py, on the box this runs on.

  python profiles/ab_bench.py [--runs 2] [--args "<bench.py arguments>"] NAME=SPEC [NAME=SPEC ...]

SPEC is a comma-separated list of `KEY=VALUE` environment settings and/or the path of a variant library (a `*.so`, e.g.
one `profiles/build_var.sh` wrote to profiles/build/); an empty SPEC is the shipped library with the default environment:

  python profiles/ab_bench.py base= depth3=REVO_TRACK_DEPTH=3 var=profiles/build/librevo_hip_var_x.so

Every run is one `python bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers <args>` in a fresh process
(the rendered inputs are cached under /tmp); the runs of the variants alternate so that box drift hits all of them alike.
Prints one line per run and the mean per variant: frames/s, ms per step, build stage alone, tracker stage alone, k_track inside
the pipelined step.  Bench arguments after a variant's `@`: NAME=SPEC@"--buffers 4 --gather-every 1".
"""
import json
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE_ARGS = "--cpu-baseline off --single-stream-frames 0 --skip-host-buffers --input-cache /tmp/revo_ab_inputs --steps 80 --warmup 8"


def parse_variant(text):
    name, _, rest = text.partition("=")
    spec, _, extra = rest.partition("@")
    env = {}
    for item in filter(None, spec.split(",")):
        if item.endswith(".so"):
            env["REVO_HIP_SO"] = os.path.abspath(item)
        else:
            k, _, v = item.partition("=")
            env[k] = v
    return name, env, shlex.split(extra)


def run_once(env_extra, args):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    if p.returncode or not lines:
        return None, (p.stderr or p.stdout)[-400:]
    return json.loads(lines[-1]), ""


def main(argv):
    runs, args, variants = 2, shlex.split(BASE_ARGS), []
    it = iter(argv)
    for a in it:
        if a == "--runs":
            runs = int(next(it))
        elif a == "--args":
            args = shlex.split(BASE_ARGS) + shlex.split(next(it))
        else:
            variants.append(parse_variant(a))
    if not variants:
        print(__doc__)
        return 2
    rows = {name: [] for name, _, _ in variants}
    print("%-14s %9s %8s %9s %9s %9s" % ("variant", "frames/s", "ms/step", "build", "tracker", "k_track"))
    for r in range(runs):
        for name, env, extra in variants:
            d, err = run_once(env, args + extra)
            if d is None:
                print("%-14s FAILED: %s" % (name, err.replace("\n", " | ")))
                continue
            row = (d["value"], d["ms_per_step"], d["stages_ms"]["pyramids_and_keyframes"], d["stages_ms"]["tracker"],
                   d["roofline"].get("kernel_ms") or float("nan"))
            rows[name].append(row)
            print("%-14s %9.0f %8.4f %9.4f %9.4f %9.4f" % ((name,) + row), flush=True)
    print("-- mean of %d run(s)" % runs)
    for name, _, _ in variants:
        if rows[name]:
            m = [sum(c) / len(c) for c in zip(*rows[name])]
            print("%-14s %9.0f %8.4f %9.4f %9.4f %9.4f" % ((name,) + tuple(m)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
