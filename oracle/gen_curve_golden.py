"""Writes tests/golden/curve_<tag>.json: the fp32 oracle's loss curves that tests/test_hip_modules.py's curve tests compare the HIP
path with (oracle/curve.py computes them; nothing of the reference is read).  CPU only; ~1 min (swin_t), ~1 min (fiber_base),
~1.5 min (swin_t_b32) on 8 cores.

    python -m oracle.gen_curve_golden [tag ...]
"""
import json
import os
import sys
import time

from . import cases, curve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tags):
    if not tags or "tiny" in tags:
        t0 = time.time()
        losses = curve.oracle_curve_tiny()
        path = os.path.join(ROOT, "tests", "golden", "curve_tiny.json")
        json.dump({"spec": curve.TINY_SPEC, "oracle": losses}, open(path, "w"), indent=0)
        print(f"tiny: {len(losses)} steps in {time.time() - t0:.0f} s, loss {losses[0]:.5f} -> {losses[-1]:.5f} -> {path}", flush=True)
        tags = [t for t in tags if t != "tiny"]
        if not tags and len(sys.argv) > 1:
            return
    for tag in tags or list(curve.CURVES):
        name, size, batch, steps, nb, warm = curve.CURVES[tag]
        t0 = time.time()
        losses = curve.oracle_curve(dict(getattr(cases, name)), size, batch, steps, nb, warm)
        out = {"spec": curve.spec_of(tag), "oracle": losses}
        path = os.path.join(ROOT, "tests", "golden", f"curve_{tag}.json")
        json.dump(out, open(path, "w"), indent=0)
        print(f"{tag}: {steps} steps in {time.time() - t0:.0f} s, loss {losses[0]:.5f} -> {losses[-1]:.5f} -> {path}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
