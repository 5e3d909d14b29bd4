#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in libfigdraw_hip.so (from the code object's metadata notes).
usage: kernel_regs.py [library.so]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lint_isa import code_objects
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "figdraw_amd", "libfigdraw_hip.so")
for triple, obj in code_objects(open(path, "rb").read()):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(obj); f.flush()
        text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    for line in text.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s*(.+)", line)
        if not m: continue
        k, v = m.groups()
        if k == "name" and v.startswith("_Z") is False and not v.startswith("_"):
            continue
        cur[k] = v
        if k == "vgpr_spill_count":
            name = subprocess.run(["c++filt", cur.get("name", "?")], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"{name:60s} vgpr {cur.get('vgpr_count'):>4} agpr {cur.get('agpr_count','0'):>3} sgpr {cur.get('sgpr_count'):>4} spills v{cur.get('vgpr_spill_count')} s{cur.get('sgpr_spill_count')} scratch {cur.get('private_segment_fixed_size')} lds {cur.get('group_segment_fixed_size')}")
            cur = {}
