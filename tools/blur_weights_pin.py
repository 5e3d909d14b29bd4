#!/usr/bin/env python3
"""One f16 per tap against hi + lo, measured: the matrix-pipe blur kernels of the product build (FDH_MX_LO = 0, 11-bit weights with the
rounding error carried outwards) and of `make variant NAME=mxlo DEFS=-DFDH_MX_LO=1` (22-bit weights, two MFMAs per operand: rounds 2 - 4)
on the hostile content of tests/ref_scenes.py (opaque white noise, a 0 / 255 checkerboard of 1-px cells; 1024 x 512, radii 5 / 18 / 64),
both routes, against (a) the reference's blur.frag on SwiftShader (tests/golden/ss_blur_big_*.png; glsl/blur.frag:11-32) and (b) the oracle.

    python tools/blur_weights_pin.py            # builds the variant if build/libfigdraw_hip_mxlo.so is missing; one child process per library
    python tools/blur_weights_pin.py --child    # (internal) the measurement for the library FIGDRAW_HIP_LIB names

The output is committed as profiles/r06_blur_weights_pin.txt.  (The oracle is the checker here, as in the tests.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    import numpy as np

    import ref_scenes as RS
    from conftest import diff_stats, load_png
    from figdraw_amd.context import HipContext
    from oracle import oracle as O

    w, h = RS.HOSTILE_BLUR_SIZE
    rows = []
    ctx = HipContext(device=0, atlas_size=2048)
    for kind in RS.HOSTILE_BLUR_KINDS:
        src = RS.hostile_blur_source(kind)
        ctx.put_image(RS.HOSTILE_BLUR_KEY, src) if not ctx.has_image(RS.HOSTILE_BLUR_KEY) else ctx.update_image(RS.HOSTILE_BLUR_KEY, src)
        for radius in RS.HOSTILE_BLUR_RADII:
            gold = load_png(f"ss_blur_big_{kind}_r{radius:g}.png")
            want = O.blur_image(src, radius)
            o_vs_g = diff_stats(want, gold)
            for route in (1, 0):
                ctx.set_blur_route(route)
                ctx.render_frame(RS.hostile_blur_scene(radius), w, h)
                got = ctx.read_pixels()
                ctx.profile(1)
                st = ctx.frame_stats()
                kernel = "k_blur_fx" if st.ms_blur_fused > 0 else "k_blur_mx H+V" if st.ms_blur_big_h > 0 else "OTHER"
                rows.append({"kind": kind, "radius": radius, "route": route, "kernel": kernel, "vs_golden": [int(v) for v in diff_stats(got, gold)],
                             "vs_oracle": [int(v) for v in diff_stats(got, want)], "oracle_vs_golden": [int(v) for v in o_vs_g]})
    ctx.close()
    print("ROWS " + json.dumps(rows))


def main():
    variant = os.path.join(ROOT, "build", "libfigdraw_hip_mxlo.so")
    if not os.path.exists(variant):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "figdraw_amd", "csrc"), "-s", "variant", "NAME=mxlo", "DEFS=-DFDH_MX_LO=1"])
    w, h = 1024, 512
    print(f"# hostile content {w} x {h} = {w * h} pixels; columns: max LSB / pixels differing / pixels differing by more than 1")
    for tag, lib in (("product build: ONE f16 per tap (11 bits, error carried outwards)", None), ("variant -DFDH_MX_LO=1: hi + lo (22 bits)", variant)):
        env = dict(os.environ)
        if lib:
            env["FIGDRAW_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(tag, "FAILED", r.stderr[-2000:])
            sys.exit(1)
        rows = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("ROWS ")][-1][5:])
        print(f"\n## {tag}")
        print(f"{'content':8} {'radius':>6} {'route':14} {'vs blur.frag on SwiftShader':>30} {'vs oracle':>22} {'(oracle vs blur.frag)':>26}")
        for x in rows:
            f = lambda t: f"{t[0]} / {t[1]} / {t[2]}"
            print(f"{x['kind']:8} {x['radius']:6g} {x['kernel']:14} {f(x['vs_golden']):>30} {f(x['vs_oracle']):>22} {f(x['oracle_vs_golden']):>26}")
        worst = max(x["vs_golden"][0] for x in rows)
        print(f"worst vs the reference shader: {worst} LSB (north-star bar: 2)")
        if worst > 2:
            sys.exit(2)


if __name__ == "__main__":
    child() if sys.argv[1:] == ["--child"] else main()
