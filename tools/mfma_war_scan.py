#!/usr/bin/env python3
"""Scan gfx950 ISA (hipcc -S --cuda-device-only) for vector instructions that WRITE a VGPR an MFMA issued a few
instructions earlier READS as SrcA / SrcB.  The hazard recognizer has no rule for that write-after-read, and a build of
the folded kernel with such a write ONE instruction behind the MFMA returned corrupted accumulator blocks for ~0.5 % of
the users, differently on every launch.  Usage: python tools/mfma_war_scan.py file.s [...]
Prints, per kernel, the closest such write (in instructions); the build keeps it >= 4 (tests/test_host_cpu.py)."""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path, window=8):
    lines = open(path).read().split("\n")
    out = {}
    name, ins = None, []
    for l in lines:
        t = l.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, ins = m.group(1), []
            continue
        if name is None or not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        ins.append(t)
        if t.startswith("s_endpgm"):
            worst = None
            for n, l1 in enumerate(ins):
                if not l1.startswith("v_mfma"):
                    continue
                ops = [x.strip() for x in l1.split(None, 1)[1].split(",")]
                src = regs(ops[1]) | regs(ops[2])
                for j in range(1, window + 1):
                    if n + j >= len(ins):
                        break
                    l2 = ins[n + j]
                    if not l2.startswith("v_") or l2.startswith("v_mfma"):
                        continue                                        # only VALU writes; LDS / memory returns come much later
                    parts = l2.split(None, 1)
                    if len(parts) > 1 and regs(parts[1].split(",")[0].strip()) & src:
                        if worst is None or j < worst[0]:
                            worst = (j, l1, l2)
                        break
            out[name] = worst
            name = None
    return out


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for k, w in scan(p).items():
            print(f"{p}: {k[:70]}: " + ("no VALU write to an MFMA source within 8 instructions" if w is None else f"+{w[0]}: {w[1][:64]} | {w[2][:56]}"))
