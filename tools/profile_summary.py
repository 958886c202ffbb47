#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the small text files kept in profiles/.

    python tools/profile_summary.py <tag> <kernel_stats.csv> [<pmc counter_collection.csv> ...]
"""
import csv
import sys


def main():
    tag, stats, pmcs = sys.argv[1], sys.argv[2], sys.argv[3:]
    what = "headline config: 100000 users x 64x4 ant x 25 paths x 512 sc"
    if pmcs and not pmcs[0].endswith(".csv"):                      # optional description of the profiled bench command
        what, pmcs = pmcs[0], pmcs[1:]
    out = [f"# rocprofv3 summary '{tag}' (python3 bench.py, {what})",
           "", "## --kernel-trace --stats (dmx kernels)", "name,calls,avg_ns,min_ns,max_ns,pct"]
    for r in csv.DictReader(open(stats)):
        if "dmx::" in r["Name"]:
            out.append(f"{r['Name'].split('(')[0]},{r['Calls']},{float(r['AverageNs']):.0f},{r['MinNs']},{r['MaxNs']},{r['Percentage']}")
    out += ["", "## --pmc passes (one counter set per run; per dispatch)", "kernel,counter,value,vgpr,lds_bytes"]
    for f in pmcs:
        seen = set()
        for r in csv.DictReader(open(f)):
            if "dmx::" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                if key in seen:
                    continue
                seen.add(key)
                out.append(f"{key[0]},{key[1]},{r['Counter_Value']},{r['VGPR_Count']},{r['LDS_Block_Size']}")
    out += ["", "WRITE_SIZE / FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced",
            "read (MI355X_MICROARCH.md, HBM section): double it before comparing with a byte count."]
    print("\n".join(out))


if __name__ == "__main__":
    main()
