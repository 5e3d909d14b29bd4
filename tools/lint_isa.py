#!/usr/bin/env python3
"""Build-time check of the gfx950 code object inside libfigdraw_hip.so: no packed-FP32 VALU instruction may be in it.

Why ("the packed-FP32 misread"; tools/microbench/pk_vs_mfma.hip reproduces it without figdraw): on
MI355X a v_pk_{fma,mul,add}_f32 whose LOW half reads the HIGH register of a VGPR pair (op_sel bit set -- what hipcc emits
for every `pair * other.y` broadcast) now and then reads that operand as 0 in lanes 48-63 while another wavefront on the
same SIMD is issuing v_mfma.  figdraw's matrix-pipe blur passes of one context run beside the compositor waves of the
others, so the product is built with -packed-fp32-ops (measured: not slower) and this script keeps it that way.

Second check: k_composite_tiles<0|2|4> are compiled with -structurizecfg-skip-uniform-regions (csrc/Makefile, the FDH_TU note in
k_composite.hip), which is only sound while its draw loop nest holds no divergent branch.  The loop nest -- every backward
branch whose range holds the draw loop's s_ff1_i32_b64 -- must therefore not write the exec mask.

usage: lint_isa.py <library.so> [--allow-packed] [--allow-missing]   exit status 1 if a forbidden opcode is present, a uniform-build
kernel writes exec inside its draw loop, or one of those kernels is not in the library at all"""
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


UNIFORM_KERNELS = tuple("k_composite_tilesILi%dELb%dE" % (paths, full) for paths in (4, 0, 2) for full in (1, 0)) + ("k_composite_deepILi1EE",)
# writes of the exec mask: s_*saveexec*, any scalar instruction whose destination is exec / exec_lo / exec_hi, and the VOPC
# compares that write exec directly (v_cmpx_*)
# (s_cmp_* / s_bitcmp* name exec as a SOURCE and write SCC only -- `s_cmp_lg_u64 exec, 0` is a wave vote over a condition the
# compiler folded to true --: not writes)
EXEC_WRITE = re.compile(r"^\s*(s_\w*saveexec\w*|(?!s_cmp_|s_bitcmp)s_\w+\s+exec(_lo|_hi)?\b|v_cmpx_\w+)")


def draw_loop_body(lines):
    """lines: disassembly of one kernel.  Returns the instructions (text) of the natural loops that hold s_ff1_i32_b64, in address
    order, or None when there is no such loop."""
    ins = []  # (address, text)
    for l in lines:
        m = re.search(r"// ([0-9A-F]{12}):", l)
        if m and l.strip():
            ins.append((int(m.group(1), 16), l))
    at = {a: i for i, (a, _) in enumerate(ins)}
    succ = [[] for _ in ins]
    for i, (a, l) in enumerate(ins):
        op = l.split()[0]
        m = re.match(r"\s*(s_cbranch_\w+|s_branch)\s+(\d+)", l)
        if m:
            off = int(m.group(2))
            off -= 65536 if off >= 32768 else 0
            t = at.get(a + 4 + 4 * off)
            if t is not None:
                succ[i].append(t)
            if op != "s_branch" and i + 1 < len(ins):
                succ[i].append(i + 1)
        elif op != "s_endpgm" and i + 1 < len(ins):
            succ[i].append(i + 1)
    # basic blocks
    leader = {0}
    for i, ss in enumerate(succ):
        if len(ss) != 1 or ss[0] != i + 1:
            leader.update(ss)
            if i + 1 < len(ins):
                leader.add(i + 1)
    starts = sorted(leader)
    blk_of = {}
    blocks = []
    for k, st in enumerate(starts):
        en = starts[k + 1] if k + 1 < len(starts) else len(ins)
        blocks.append((st, en))
        for i in range(st, en):
            blk_of[i] = k
    bsucc = [sorted({blk_of[t] for t in succ[en - 1]}) for st, en in blocks]
    bpred = [[] for _ in blocks]
    for k, ss in enumerate(bsucc):
        for t in ss:
            bpred[t].append(k)
    # dominators (iterative, bit sets)
    n = len(blocks)
    full = (1 << n) - 1
    dom = [full] * n
    dom[0] = 1
    changed = True
    while changed:
        changed = False
        for k in range(1, n):
            d = full
            for q in bpred[k]:
                d &= dom[q]
            d |= 1 << k
            if d != dom[k]:
                dom[k] = d
                changed = True
    marks = {blk_of[i] for i, (_, l) in enumerate(ins) if "s_ff1_i32_b64" in l}
    if not marks:
        return None
    body = set()
    for u in range(n):
        for h in bsucc[u]:
            if not (dom[u] >> h) & 1:
                continue  # not a back edge
            loop, work = {h, u}, [u]
            while work:
                x = work.pop()
                if x == h:
                    continue
                for q in bpred[x]:
                    if q not in loop:
                        loop.add(q)
                        work.append(q)
            if marks & loop:
                body |= loop
    if not body:
        return None
    out = []
    for k in sorted(body):
        out += [ins[i][1].strip() for i in range(*blocks[k])]
    return out


def exec_writes_in_draw_loop(lines):
    """Returns the offending lines inside the natural loops that hold s_ff1_i32_b64."""
    body = draw_loop_body(lines)
    if body is None:
        return ["(draw loop not found)"]
    return [l for l in body if EXEC_WRITE.search(l)]


def main():
    path = sys.argv[1]
    blob = open(path, "rb").read()
    found = False
    bad = {}
    n_inst = 0
    uniform_lines = {}
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
            for uk in UNIFORM_KERNELS:
                if uk in kernel:  # (one entry per KERNEL: a pattern matches several instantiations -- <., ., direct> -- and their lines must not be analysed as one)
                    uniform_lines.setdefault((uk, kernel), []).append(line)
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
    rc = 0
    # (a renamed or dropped kernel must not turn the check into a silent pass: the Makefile still compiles that unit with the switch)
    missing = [uk for uk in UNIFORM_KERNELS if uk not in {k[0] for k in uniform_lines}]
    if missing and "--allow-missing" not in sys.argv:
        print(f"lint_isa: kernels compiled with -structurizecfg-skip-uniform-regions not found in {path}: {missing} "
              "(update UNIFORM_KERNELS in tools/lint_isa.py when the compositor's template arguments change)")
        rc = 1
    for (uk, kernel), lines in sorted(uniform_lines.items()):
        name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void fdh::", "") or kernel
        w = exec_writes_in_draw_loop(lines)
        if w:
            print(f"lint_isa: {name} has a divergent branch inside its draw loop (it is compiled with "
                  "-structurizecfg-skip-uniform-regions, which needs that loop nest free of them):")
            for l in w[:10]:
                print("  " + l[:100])
            rc = 1
        else:
            print(f"lint_isa: {name}: no exec-mask write inside the draw loop nest")
    return rc


if __name__ == "__main__":
    sys.exit(main())
