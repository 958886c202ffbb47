#!/usr/bin/env python3
"""profiles/traffic.json from the WRITE_SIZE / FETCH_SIZE PMC passes of the stage-2 kernel.

    python tools/make_traffic_json.py <workload> <users> <variant> <write_counter_collection.csv> <fetch_counter_collection.csv> <source tag> [kernel substring]

The entry records the id of the library build it was measured on (bench.library_id(): the built .so travels with the
repo snapshot to the GPU box, so the id is the same here and there); bench.py reports the figure as `roofline.traffic`
only for that build.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def k2_value(path, counter, kernel="dmx::k2_fd"):
    """`kernel`: substring of the kernel's name; several, comma-separated, are summed (rx_filter: FFT + contraction)."""
    total, found = 0.0, 0
    for sub in kernel.split(","):
        for r in csv.DictReader(open(path)):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                total += float(r["Counter_Value"])
                found += 1
                break
    if found != len(kernel.split(",")):
        raise SystemExit(f"{counter} of the stage-2 kernel(s) '{kernel}' not found in {path}")
    return total


def main():
    workload, users, variant, wcsv, fcsv, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    kernel = sys.argv[7] if len(sys.argv) > 7 else "dmx::k2_fd"
    sys.path.insert(0, ROOT)
    import bench
    w_kib, f_kib = k2_value(wcsv, "WRITE_SIZE", kernel), k2_value(fcsv, "FETCH_SIZE", kernel)
    # matrix-core busy fraction of the same kernel when the SQ / GRBM pass is beside the TCC passes:
    # SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024 SIMDs)
    mfma_busy = None
    import glob
    for g in glob.glob(os.path.join(os.path.dirname(os.path.dirname(wcsv)).replace("_pmc_WRITE_SIZE", "_pmc_GRBM*"), "*", "*counter_collection.csv")):
        try:
            k0 = kernel.split(",")[0]
            mfma_busy = k2_value(g, "SQ_VALU_MFMA_BUSY_CYCLES", k0) / (k2_value(g, "GRBM_GUI_ACTIVE", k0) / 8 * 1024)
        except SystemExit:
            pass
    out = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(out)) if os.path.exists(out) else {}
    t[workload] = {"users": users, "variant": variant, "write_size_kib": w_kib, "fetch_size_kib": f_kib,
                   "hbm_bytes_per_launch": w_kib * 1024 + 2 * f_kib * 1024,
                   "note": "WRITE_SIZE + 2 x FETCH_SIZE (gfx950: FETCH_SIZE counts half of a wide coalesced read)",
                   "mfma_busy_frac": mfma_busy, "profile": tag, "library_id": bench.library_id()}
    json.dump(t, open(out, "w"), indent=1)
    print(json.dumps(t[workload]))


if __name__ == "__main__":
    main()
