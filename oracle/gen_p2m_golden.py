"""Golden vector for the .p2m parser (build container only): writes a SYNTHETIC Wireless InSite paths file in
the layout deepmimo/converter/wireless_insite/p2m_parser.py expects, parses it with the REAL reference parser
and stores text + expected arrays in tests/golden/p2m_paths.npz.

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 /root/repo/oracle/gen_p2m_golden.py
"""
import io
import os
import sys
import tempfile
from contextlib import redirect_stdout, redirect_stderr

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "p2m_paths.npz")


def synth_p2m(n_rx=40, seed=7) -> str:
    rng = np.random.default_rng(seed)
    lines = [f"# synthetic header line {i}" for i in range(21)]
    lines.append(str(n_rx))
    codes = ["R", "D", "DS", "T", "F", "X"]
    for rx in range(1, n_rx + 1):
        n_paths = int(rng.choice([0, 0, 1, 2, 5, 9, 17, 25]))
        lines.append(f"{rx} {n_paths}")
        if n_paths == 0:
            continue
        lines.append(f"{rng.uniform(-150, -60):.4f} {rng.uniform(1e-7, 3e-6):.6e} {rng.uniform(0, 1e-6):.6e}")   # summary line
        for p in range(1, n_paths + 1):
            n_int = int(rng.integers(0, 5))
            vals = [rng.uniform(-170, -60), rng.uniform(-180, 180), rng.uniform(1e-8, 4e-6), rng.uniform(0, 180),
                    rng.uniform(-180, 180), rng.uniform(0, 180), rng.uniform(-180, 180)]
            lines.append(f"{p} {n_int} {vals[0]:.3f} {vals[1]:.4f} {vals[2]:.7e} {vals[3]:.5f} {vals[4]:.5f} {vals[5]:.5f} {vals[6]:.5f}")
            lines.append("-".join(["Tx"] + [str(rng.choice(codes)) for _ in range(n_int)] + ["Rx"]))
            for _ in range(n_int + 2):
                x, y, z = rng.uniform(-500, 500, 3)
                lines.append(f"{x:.4f} {y:.4f} {z:.3f}")
    return "\n".join(lines) + "\n"


def main():
    sys.path.insert(0, "/root/reference")
    from deepmimo.converter.wireless_insite.p2m_parser import paths_parser
    text = synth_p2m()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synthetic.paths.t001_01.r002.p2m")
        with open(path, "w") as f:
            f.write(text)
        with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
            ref = paths_parser(path)
    save = {f"ref_{k}": v for k, v in ref.items()}
    save["p2m_text"] = np.array(text)
    np.savez_compressed(OUT, **save)
    print({k: v.shape for k, v in ref.items()}, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
