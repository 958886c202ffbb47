#!/usr/bin/env python3
"""Static checks on the gfx950 machine code of the matrix-core kernels (no GPU needed).

Reads the disassembly of a code object (`llvm-objdump -d`; the code objects are pulled out of the built shared library
with `llvm-objdump --offloading`, or a device-only object from `hipcc --cuda-device-only -c` is given directly) and, for
every `v_mfma_*` instruction of every kernel, walks the control-flow graph forward and measures in WAIT STATES (one per
issued instruction, n + 1 for `s_nop n`):

  raw_lo / raw_hi   first non-MFMA read of the MFMA's result registers (first / second half of vDst - the second half is
                    written by the last passes) counted from the LAST MFMA that writes them.  The hazard table needs 12
                    for the 8-pass 32x32x16 f16 instruction on gfx950 (profiles/r2_mfma_hazard_probes.txt item 3 measures
                    the same 12); the kernels add a guard of 4 on top.
  ret_X             a DS / VMEM *load* whose destination registers overlap operand X (A, B, C = SrcC/vDst) of an MFMA
                    issued at most `window` wait states earlier - an asynchronous return landing in a register an
                    in-flight MFMA names.  No hazard-recognizer rule covers it; the kernels are written so that it does
                    not occur (operands stay live across the next loads).  `pipe` distance counts 8 wait states for
                    every MFMA (a back-to-back group of n occupies the matrix pipe for 8 n).
  valu_X            the same for an ordinary vector instruction writing such a register (hazard table: 12 / 13 wait
                    states for vDst / SrcC; nothing for A / B, which are read while the MFMA holds the issue port).
  dep               distance from an MFMA to the next MFMA that accumulates onto its result, when not back to back.

Used by tests/test_isa_lint.py (the lint on the shipped library) and by tools/isa_incident_table.py (the table of the
round-2 folded-kernel builds in DESIGN.md section 4).
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
from dataclasses import dataclass, field

LLVM_BIN = os.environ.get("DMX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
OBJDUMP = os.path.join(LLVM_BIN, "llvm-objdump")

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def _regs(tok: str) -> set:
    """VGPR / AGPR numbers named by one operand token (AGPRs are numbered 1000+)."""
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            base = 1000 if m.group(1) == "a" else 0
            out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
        else:
            base = 1000 if m.group(4) == "a" else 0
            out.add(base + int(m.group(5)))
    return out


@dataclass
class Inst:
    addr: int
    text: str
    mnem: str
    kind: str                      # mfma | valu | ds_load | ds_other | vm_load | vm_other | nop | branch | cbranch | end | other
    defs: frozenset = frozenset()
    uses: frozenset = frozenset()
    ws: int = 1                    # wait states this instruction occupies at issue
    target: int = -1               # branch target address
    opA: frozenset = frozenset()   # MFMA operands
    opB: frozenset = frozenset()
    opC: frozenset = frozenset()
    dst_order: tuple = ()          # MFMA vDst registers in order


_NO_VGPR_DST = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")
_DST_ALSO_SRC = ("v_fmac", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c", "v_pk_fmac")
_TWO_DST = ("v_swap_b32", "v_permlane16_swap", "v_permlane32_swap")


def _split_ops(s: str):
    ops, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def parse_inst(addr: int, text: str) -> Inst:
    parts = text.split(None, 1)
    mnem = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    # modifiers (op_sel:[..], offset:.., offen ...) carry no registers except in DPP/SDWA forms, which name none either
    ops = _split_ops(rest)
    ops_regs = [_regs(o.split(" ")[0]) if o else set() for o in ops]
    # operands followed by modifiers: "v76 offset:256" -> first word only; keep full scan for safety on the last operand
    if ops:
        ops_regs[-1] = _regs(ops[-1].split(" ")[0])
    ins = Inst(addr, text, mnem, "other")
    if mnem.startswith("v_mfma") or mnem.startswith("v_smfmac"):
        ins.kind = "mfma"
        d = ops_regs[0]
        ins.defs = frozenset(d)
        ins.dst_order = tuple(sorted(d))
        ins.opA, ins.opB = frozenset(ops_regs[1]), frozenset(ops_regs[2])
        ins.opC = frozenset(ops_regs[3]) if len(ops_regs) > 3 else frozenset()
        ins.uses = frozenset(ins.opA | ins.opB | ins.opC)
    elif mnem == "s_nop":
        ins.kind = "nop"
        ins.ws = int(rest.strip(), 0) + 1
    elif mnem == "s_endpgm":
        ins.kind = "end"
    elif mnem == "s_branch":
        ins.kind = "branch"
    elif mnem.startswith("s_cbranch"):
        ins.kind = "cbranch"
    elif mnem.startswith("v_"):
        ins.kind = "valu"
        if mnem.startswith(_NO_VGPR_DST):
            ins.uses = frozenset(set().union(*ops_regs)) if ops_regs else frozenset()
        elif mnem.startswith(_TWO_DST):
            both = ops_regs[0] | ops_regs[1]
            ins.defs, ins.uses = frozenset(both), frozenset(both)
        else:
            ins.defs = frozenset(ops_regs[0]) if ops_regs else frozenset()
            u = set().union(*ops_regs[1:]) if len(ops_regs) > 1 else set()
            if mnem.startswith(_DST_ALSO_SRC):
                u |= ops_regs[0]
            ins.uses = frozenset(u)
    elif mnem.startswith("ds_"):
        is_load = mnem.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle")) or "_rtn" in mnem or "ret" in mnem
        if is_load:
            ins.kind = "ds_load"
            ins.defs = frozenset(ops_regs[0]) if ops_regs else frozenset()
            ins.uses = frozenset(set().union(*ops_regs[1:])) if len(ops_regs) > 1 else frozenset()
        else:
            ins.kind = "ds_other"
            ins.uses = frozenset(set().union(*ops_regs)) if ops_regs else frozenset()
    elif mnem.startswith(("buffer_", "global_", "flat_", "scratch_")):
        is_load = "_load" in mnem or ("atomic" in mnem and (" sc0" in text or " glc" in text))
        to_lds = " lds" in text
        if is_load and not to_lds:
            ins.kind = "vm_load"
            ins.defs = frozenset(ops_regs[0]) if ops_regs else frozenset()
            ins.uses = frozenset(set().union(*ops_regs[1:])) if len(ops_regs) > 1 else frozenset()
        else:
            ins.kind = "vm_other"
            ins.uses = frozenset(set().union(*ops_regs)) if ops_regs else frozenset()
    return ins


_LINE = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
_BR = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(-?\d+)")


def parse_objdump(text: str) -> dict:
    """llvm-objdump -d output -> {kernel symbol: [Inst]} with branch targets resolved to addresses."""
    kernels, cur = {}, None
    for line in text.split("\n"):
        m = _FUNC.match(line)
        if m:
            cur = []
            kernels[m.group(1)] = cur
            continue
        m = _LINE.match(line)
        if not m or cur is None:
            continue
        addr = int(m.group(2), 16)
        ins = parse_inst(addr, m.group(1))
        b = _BR.match(ins.text)
        if b:
            simm = int(b.group(2))
            if simm >= 32768:
                simm -= 65536
            ins.target = addr + 4 + 4 * simm
        cur.append(ins)
    return {k: v for k, v in kernels.items() if v}


def disassemble(path: str) -> str:
    return subprocess.run([OBJDUMP, "-d", path], check=True, capture_output=True, text=True).stdout


def extract_code_objects(shared_lib: str, workdir: str) -> list:
    """gfx950 code objects bundled in a host shared library (one per translation unit)."""
    tmp = os.path.join(workdir, os.path.basename(shared_lib))
    shutil.copy(shared_lib, tmp)
    subprocess.run([OBJDUMP, "--offloading", tmp], check=True, capture_output=True, text=True, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "gfx950" in f and not f.endswith(".s"))


def kernels_of_library(shared_lib: str) -> dict:
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for co in extract_code_objects(shared_lib, d):
            out.update(parse_objdump(disassemble(co)))
    return out


# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class MfmaReport:
    kernel: str
    n_mfma: int = 0
    raw_lo: int = 10**9            # min wait states, last MFMA -> first non-MFMA read of vDst[0 : n/2]
    raw_hi: int = 10**9            # ... of vDst[n/2 : n]
    raw_lo_at: str = ""
    raw_hi_at: str = ""
    ret: dict = field(default_factory=dict)    # operand -> (min issue distance, min pipe distance, count, example)
    valu: dict = field(default_factory=dict)   # operand -> (min distance, count, example)
    dep_min: int = 10**9           # non-back-to-back dependent MFMA distances
    dep_max: int = 0
    dep_hist: dict = field(default_factory=dict)
    a_is_result: int = 0           # MFMAs reading another MFMA's fresh result as A/B within the window (needs nops)
    dep_branches: int = 0          # dependent-MFMA pairs with a branch instruction between them (control flow inside a chain)


def analyse_kernel(name: str, insts: list, window: int = 40) -> MfmaReport:
    rep = MfmaReport(name)
    index = {ins.addr: i for i, ins in enumerate(insts)}
    n = len(insts)

    def succ(i):
        ins = insts[i]
        if ins.kind == "end":
            return []
        if ins.kind == "branch":
            return [index[ins.target]] if ins.target in index else []
        if ins.kind == "cbranch":
            s = [i + 1] if i + 1 < n else []
            if ins.target in index:
                s.append(index[ins.target])
            return s
        return [i + 1] if i + 1 < n else []

    for i, mf in enumerate(insts):
        if mf.kind != "mfma":
            continue
        rep.n_mfma += 1
        D = mf.defs
        order = mf.dst_order
        half = len(order) // 2
        lo, hi = frozenset(order[:half]), frozenset(order[half:])
        ops = {"A": mf.opA, "B": mf.opB, "C": D | mf.opC}
        # DFS over (instruction, wait states BETWEEN the MFMA and it - 0 = back to back, the hazard tables' convention -,
        # the same in pipe time, registers of D this MFMA is still the last writer of)
        best = {}
        stack = [(j, 0, 7, D, 0) for j in succ(i)]
        while stack:
            j, ws, pipe, live, nbr = stack.pop()
            if ws > window:
                continue
            key = (j, live, nbr > 0)
            if key in best and best[key] <= ws:
                continue
            best[key] = ws
            ins = insts[j]
            if ins.kind == "mfma":
                if ins.opC and ins.opC & D and live:
                    d = ws
                    if d > 0:                                  # not back to back
                        rep.dep_min = min(rep.dep_min, d)
                        rep.dep_max = max(rep.dep_max, d)
                        b = min(d // 8 * 8, 64)
                        rep.dep_hist[b] = rep.dep_hist.get(b, 0) + 1
                    if nbr:
                        rep.dep_branches += 1
                if (ins.opA | ins.opB) & live:
                    rep.a_is_result += 1
                if ins.defs & live:
                    live = live - ins.defs                     # a later MFMA writes D: it is measured on its own
            else:
                if ins.uses & live:
                    if ins.uses & lo and ws < rep.raw_lo:
                        rep.raw_lo, rep.raw_lo_at = ws, f"{mf.addr:x}->{ins.addr:x} {ins.text[:48]}"
                    if ins.uses & hi and ws < rep.raw_hi:
                        rep.raw_hi, rep.raw_hi_at = ws, f"{mf.addr:x}->{ins.addr:x} {ins.text[:48]}"
                if ins.defs:
                    for opn, regs in ops.items():
                        if not (ins.defs & regs):
                            continue
                        if opn == "C" and not (ins.defs & live):
                            continue                           # belongs to a later writer of D (measured there)
                        if ins.kind in ("ds_load", "vm_load"):
                            o = rep.ret.get(opn, (10**9, 10**9, 0, ""))
                            ex = o[3] if o[0] <= ws else f"{mf.addr:x}->{ins.addr:x} {ins.text[:44]}"
                            rep.ret[opn] = (min(o[0], ws), min(o[1], pipe), o[2] + 1, ex)
                        elif ins.kind == "valu":
                            o = rep.valu.get(opn, (10**9, 0, ""))
                            ex = o[2] if o[0] <= ws else f"{mf.addr:x}->{ins.addr:x} {ins.text[:44]}"
                            rep.valu[opn] = (min(o[0], ws), o[1] + 1, ex)
                    live = live - ins.defs                     # overwritten: no longer this MFMA's result
            add_ws = ins.ws
            add_pipe = 8 if ins.kind == "mfma" else ins.ws
            nb2 = nbr + (1 if ins.kind in ("branch", "cbranch") else 0)
            for k in succ(j):
                stack.append((k, ws + add_ws, pipe + add_pipe, live, nb2))
    return rep


def analyse(kernels: dict, window: int = 40) -> dict:
    return {k: analyse_kernel(k, v, window) for k, v in kernels.items() if any(i.kind == "mfma" for i in v)}


def short_name(sym: str) -> str:
    try:
        d = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    except OSError:
        d = sym
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*$", "", d)
    return d.replace("dmx::", "")


def format_report(reps: dict) -> str:
    lines = []
    for k, r in sorted(reps.items(), key=lambda kv: short_name(kv[0])):
        lines.append(f"{short_name(k)}: {r.n_mfma} MFMAs")
        lines.append(f"    result read   first half >= {r.raw_lo if r.raw_lo < 10**9 else '-'} ws   second half >= {r.raw_hi if r.raw_hi < 10**9 else '-'} ws"
                     f"   [{r.raw_lo_at}]")
        for opn in ("A", "B", "C"):
            if opn in r.ret:
                d, p, c, ex = r.ret[opn]
                lines.append(f"    load return into operand {opn}: closest {d} ws behind the MFMA ({p} in pipe time), {c} sites   [{ex}]")
        for opn in ("A", "B", "C"):
            if opn in r.valu:
                d, c, ex = r.valu[opn]
                lines.append(f"    vector write into operand {opn}: closest {d} ws behind the MFMA, {c} sites   [{ex}]")
        if r.dep_max:
            lines.append(f"    dependent MFMA not back to back: {r.dep_min}..{r.dep_max} ws; histogram by 8 ws {dict(sorted(r.dep_hist.items()))}")
        if r.dep_branches:
            lines.append(f"    dependent MFMA pairs with a branch between them: {r.dep_branches}")
        if r.a_is_result:
            lines.append(f"    MFMA reading a fresh MFMA result as A/B: {r.a_is_result}")
    return "\n".join(lines)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("path", nargs="+", help="shared library, gfx950 code object / device-only object, or objdump text (.s)")
    ap.add_argument("--window", type=int, default=40)
    a = ap.parse_args()
    for p in a.path:
        if p.endswith(".so"):
            ks = kernels_of_library(p)
        elif p.endswith(".s"):
            ks = parse_objdump(open(p).read())
        else:
            ks = parse_objdump(disassemble(p))
        print(f"== {p}")
        print(format_report(analyse(ks, a.window)))
