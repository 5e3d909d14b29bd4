#!/usr/bin/env python3
"""Build-time check of the gfx950 code object inside libfigdraw_hip.so: no packed-FP32 VALU instruction may be in it.

Why ("the packed-FP32 misread"; tools/microbench/pk_vs_mfma.hip reproduces it without figdraw): on
MI355X a v_pk_{fma,mul,add}_f32 whose LOW half reads the HIGH register of a VGPR pair (op_sel bit set -- what hipcc emits
for every `pair * other.y` broadcast) now and then reads that operand as 0 in lanes 48-63 while another wavefront on the
same SIMD is issuing v_mfma.  figdraw's matrix-pipe blur passes of one context run beside the compositor waves of the
others, so the product is built with -packed-fp32-ops (measured: not slower) and this script keeps it that way.

usage: lint_isa.py <library.so> [--allow-packed]   exit status 1 if a forbidden opcode is present"""
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FORBIDDEN = re.compile(r"\bv_pk_(fma|mul|add)_f32\b|\bv_pk_mov_b32\b")


def code_objects(blob: bytes):
    """yield (triple, bytes) of every device entry of every uncompressed offload bundle in the file"""
    at = 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return
        n = struct.unpack_from("<Q", blob, at + len(MAGIC))[0]
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if size and "amdgcn" in triple:
                yield triple, blob[at + off:at + off + size]
        at += len(MAGIC)


def main():
    path = sys.argv[1]
    blob = open(path, "rb").read()
    found = False
    bad = {}
    n_inst = 0
    for triple, obj in code_objects(blob):
        found = True
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
                continue
            n_inst += 1
            m = FORBIDDEN.search(line)
            if m:
                bad.setdefault(kernel, {}).setdefault(m.group(0), 0)
                bad[kernel][m.group(0)] += 1
    if not found:
        print(f"lint_isa: no gfx code object found in {path} (compressed bundle?)")
        return 1
    if bad and "--allow-packed" not in sys.argv:
        print(f"lint_isa: packed-FP32 instructions in {path} (build the device side with -packed-fp32-ops):")
        for k, v in sorted(bad.items()):
            print(f"  {k[:100]}: {v}")
        return 1
    print(f"lint_isa: {path}: {n_inst} lines of gfx950 disassembly, packed-FP32 instructions: {sum(sum(v.values()) for v in bad.values())}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
