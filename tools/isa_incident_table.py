#!/usr/bin/env python3
"""Machine-code table of the round-2 folded-kernel builds that were (not) bit-reproducible.

For every build - a commit of this repository, compiled here with the flags of deepmimo_amd/csrc/Makefile; no GPU - the
per-kernel figures of tools/isa_lint.py next to the failure rate the round-2 records hold for it, plus the plain
matrix-core kernel of the same commit as the control (it never differed in 1.75e8 user-launches = 4.5e10 tiles on any
box, susceptible ones included).  `python tools/isa_incident_table.py > profiles/r3_fold_incident_isa_table.txt`
"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_lint  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# commit, what the tile loop looked like, recorded rate (DESIGN.md section 4 of round 2, profiles/r2_mfma_hazard_probes.txt,
# commit messages 82cd286 / 2275029 / b313e09 / 288c216 / 3b4e843)
BUILDS = [
    ("82cd286^", "before the adaptive rule: run-time guarded K-steps, v_fma_mix as inline asm", "never stress-tested (no reproducibility test existed yet)"),
    ("82cd286", "per-tile switch over templated bodies (merged by the compiler into shared blocks), one accumulator", "1 user-launch in 1e7, on EVERY box (12-15 per 1e8)"),
    ("2275029", "(K-steps, weak tail) chosen outside the tile loop, straight-line bodies, one accumulator", "1 in 1e7 ... 1e9 on SOME boxes, 0 in 2.4e8 on the others"),
    ("b313e09", "same + operand-less `s_nop 3` behind the last MFMA (in front of the reads in THIS build)", "14 in 6e8 (2.3e-8) on a susceptible box, 0 in 3e9 elsewhere"),
    ("288c216", "two accumulators (even / odd K-steps), 4 waves per SIMD; = round-2 HEAD", "0 in 1.4e9 on the same susceptible box"),
    ("WORKTREE", "round 3: operands of the asm guard, table entries in registers of their own, scalar row offsets", "(this build)"),
]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "--no-gpu-bundle-output", "-c"]


def build(commit, src, workdir):
    d = os.path.join(workdir, commit.replace("^", "_parent"))
    os.makedirs(d, exist_ok=True)
    if commit == "WORKTREE":
        srcdir = os.path.join(ROOT, "deepmimo_amd", "csrc")
    else:
        tar = subprocess.run(["git", "-C", ROOT, "archive", commit, "deepmimo_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", d], input=tar, check=True)
        srcdir = os.path.join(d, "deepmimo_amd", "csrc")
    obj = os.path.join(d, src.replace(".hip", ".o"))
    subprocess.run(["hipcc"] + FLAGS + [src, "-o", obj], cwd=srcdir, check=True, capture_output=True)
    notes = subprocess.run([os.path.join(isa_lint.LLVM_BIN, "llvm-readelf"), "--notes", obj], capture_output=True, text=True).stdout
    vg, name = {}, None
    for line in notes.split("\n"):
        t = line.strip()
        if t.startswith(".name:"):
            name = t.split()[-1]
        elif t.startswith(".vgpr_count:") and name:
            vg[name] = int(t.split()[-1])
    return isa_lint.analyse(isa_lint.parse_objdump(isa_lint.disassemble(obj)), window=60), vg


def fmt(x):
    return "-" if x >= 10**9 else str(x)


def row(name, r, vgpr):
    waves = 512 // ((vgpr + 7) // 8 * 8) if vgpr else 0
    ra, rc = r.ret.get("A"), r.ret.get("C")
    va, vc = r.valu.get("A"), r.valu.get("C")
    dep = f"{r.dep_min}..{r.dep_max}" if r.dep_max else "none in 60"
    return (f"  {name:34s} vgpr {vgpr:3d} -> {min(waves, 8)} waves/SIMD | result read >= {fmt(r.raw_lo)} / {fmt(r.raw_hi)} ws (regs 0-7 / 8-15) | "
            f"load -> A {fmt(ra[0]) if ra else '-'}  load -> C {fmt(rc[0]) if rc else '-'} | valu -> A {fmt(va[0]) if va else '-'}  valu -> C {fmt(vc[0]) if vc else '-'} | "
            f"dependent MFMA apart: {dep} ws, branch inside a chain: {r.dep_branches}")


def main():
    print(__doc__)
    print("columns: allocated VGPRs -> waves per SIMD | wait states from the LAST MFMA writing an accumulator to the first vector read of its\n"
          "first / second half | closest DS/VMEM load whose destination is operand A / the accumulator of an MFMA issued that many wait\n"
          "states earlier | the same for a vector instruction | distance between dependent MFMAs that are not back to back, and how many\n"
          "such pairs have a branch between them.  (0 = the very next instruction; '-' = none within 60 wait states.)\n")
    with tempfile.TemporaryDirectory() as wd:
        for commit, what, rate in BUILDS:
            print(f"{commit}: {what}\n    recorded: {rate}")
            reps, vg = build(commit, "k2_channel_fd_fold.hip", wd)
            for k, r in sorted(reps.items()):
                print(row(isa_lint.short_name(k), r, vg.get(k, 0)))
            if commit in ("2275029", "WORKTREE"):
                reps, vg = build(commit, "k2_channel_fd_mfma.hip", wd)
                print("    control - plain matrix-core kernel of the same commit (never differed, 4.5e10 tiles, every box):")
                for k, r in sorted(reps.items()):
                    n = isa_lint.short_name(k)
                    if n.startswith("k2_fd_mfma<true, 8,") and n.endswith(", 0>"):
                        print(row(n, r, vg.get(k, 0)))
            print()


if __name__ == "__main__":
    main()
