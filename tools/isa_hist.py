#!/usr/bin/env python3
"""Static opcode histogram of one kernel of libfigdraw_hip.so.
usage: isa_hist.py <substring of the mangled name> [library] [--dump] [--loop] [--classes]
  --loop      only the natural loops around the compositor's draw loop (the ones that hold s_ff1_i32_b64; tools/lint_isa.py's notion)
  --classes   VALU instructions by issue class ("What the VALU gives"; tools/microbench/valu_rate.hip):
                fma      2.4 cycles: v_fma / v_fmac / v_mul / v_add / v_sub, moves, integer and logic ops, all operands VGPRs or constants
                sgpr     the same opcodes with an SGPR source: 4 cycles
                select   4 cycles by opcode: v_min / v_max / v_med3 / v_cmp / v_cndmask / v_rndne / v_floor / conversions / lane reads
                trans    8 cycles: v_exp / v_rcp / v_rsq / v_sqrt / v_log
              (static counts: what the draw loop holds, not what a frame executes -- tools/pmc_classes.sh counts that)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lint_isa import code_objects, draw_loop_body  # noqa: E402

TRANS = re.compile(r"^v_(exp|rcp|rsq|sqrt|log|sin|cos)_")
SELECT = re.compile(r"^v_(min|max|med3|cmp|cmpx|cndmask|rndne|floor|ceil|trunc|fract|cvt|readlane|readfirstlane|writelane|perm|bfe|bfi|alignbit|sad|mbcnt)")
SGPR_SRC = re.compile(r"\b(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|ttmp\d+|m0|exec(_lo|_hi)?)\b")


def valu_class(line):
    """class of one disassembled VALU instruction"""
    text = line.split("//")[0]
    op, _, rest = text.strip().partition(" ")
    if TRANS.match(op):
        return "trans"
    if SELECT.match(op):
        return "select"
    ops = [o.strip() for o in rest.split(",")]
    srcs = ops[1:]
    if op.startswith("v_cndmask") or op.endswith("_e64") and op.startswith("v_cmp"):
        srcs = ops[1:]
    # carry-out / carry-in operands (vcc or an SGPR pair) of v_add_co / v_addc are not data sources read through the constant port
    if re.match(r"^v_(add|sub|subrev)(c)?_co_", op):
        srcs = [o for o in srcs if not re.fullmatch(r"vcc|s\[\d+:\d+\]", o)]
    return "sgpr" if any(SGPR_SRC.search(o) for o in srcs) else "fma"


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = args[0]
    path = args[1] if len(args) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "figdraw_amd", "libfigdraw_hip.so")
    for triple, obj in code_objects(open(path, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout
        cur = None
        lines = []
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                continue
            if cur and pat in cur and line.strip():
                lines.append(line)
        if not lines:
            continue
        if "--loop" in flags:
            lines = draw_loop_body(lines) or []
        if "--dump" in flags:
            print("\n".join(lines))
            continue
        hist = collections.Counter(l.split()[0] for l in lines)
        tot = sum(hist.values())
        nv = sum(v for k, v in hist.items() if k.startswith("v_"))
        print(f"{tot} instructions{' in the draw loop nest' if '--loop' in flags else ''}; VALU {nv}  SALU {sum(v for k, v in hist.items() if k.startswith('s_'))}")
        if "--classes" in flags:
            cls = collections.Counter()
            per = collections.defaultdict(collections.Counter)
            for l in lines:
                op = l.split()[0]
                if op.startswith("v_"):
                    c = valu_class(l)
                    cls[c] += 1
                    per[c][op] += 1
            for c in ("fma", "sgpr", "select", "trans"):
                top = ", ".join(f"{k} {v}" for k, v in per[c].most_common(8))
                print(f"  {c:7s} {cls[c]:5d}  {100.0 * cls[c] / max(nv, 1):5.1f} %   {top}")
            cyc = 2.4 * cls["fma"] + 4.0 * (cls["sgpr"] + cls["select"]) + 8.0 * cls["trans"]
            print(f"  issue cycles if every instruction paid its class: {cyc:.0f} ({cyc / max(nv, 1):.2f} per VALU instruction); 4-cycle classes {100.0 * (cls['sgpr'] + cls['select']) / max(nv, 1):.1f} % of the VALU instructions")
        else:
            for k, v in hist.most_common(45):
                print(f"{k:28s} {v}")


if __name__ == "__main__":
    main()
