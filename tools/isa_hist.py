#!/usr/bin/env python3
"""Static opcode histogram of one kernel of libfigdraw_hip.so.  usage: isa_hist.py <substring of the mangled name> [library] [--dump]"""
import os, re, subprocess, sys, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lint_isa import code_objects
pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "figdraw_amd", "libfigdraw_hip.so")
for triple, obj in code_objects(open(path, "rb").read()):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(obj); f.flush()
        text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout
    cur = None; hist = collections.Counter(); lines = []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1); continue
        if cur and pat in cur and line.strip():
            op = line.split()[0]
            hist[op] += 1; lines.append(line)
    if "--dump" in sys.argv:
        print("\n".join(lines))
    else:
        tot = sum(hist.values())
        print(f"{tot} instructions; VALU {sum(v for k, v in hist.items() if k.startswith('v_'))}  SALU {sum(v for k, v in hist.items() if k.startswith('s_'))}")
        for k, v in hist.most_common(45):
            print(f"{k:28s} {v}")
