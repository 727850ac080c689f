"""Summarise two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/rNN_hbm_pmc.csv.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py <args>
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py <args>
    python tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w CHAINS "<args>" > profiles/r01_hbm_pmc.csv

CHAINS ("auto" = the number of otsu_f64_kernel launches) = number of executions of the 32-FOV chain in the profiled run (profile pass + warm-up + steps); every
value is divided by it, so a row is "per chain execution".  rocprofv3 reports both counters in KiB.  On gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes (same guide), hence the fetch_x2
column; bench.py uses fetch_x2 + write as the stage's HBM traffic.
"""
from __future__ import annotations

import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    return name.split("(")[0].strip()


def collect(directory: str, counter: str):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    tot, calls = defaultdict(float), defaultdict(int)
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row["Kernel_Name"])
                tot[k] += float(row["Counter_Value"])
                calls[k] += 1
    return tot, calls


def main():
    fdir, wdir, args = sys.argv[1], sys.argv[2], sys.argv[4]
    fetch, calls = collect(fdir, "FETCH_SIZE")
    # "auto": the Otsu kernel runs exactly once per chain execution
    chains = float(calls["otsu_f64_kernel"]) if sys.argv[3] == "auto" else float(sys.argv[3])
    write, _ = collect(wdir, "WRITE_SIZE")
    kib = 1024.0 / 1e6  # KiB -> MB
    print(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py {args}")
    print(f"# values are per chain execution over a batch of 32 FOVs ({chains:g} chain executions in the run); "
          "FETCH_SIZE/WRITE_SIZE are in KiB (rocprofv3 units);")
    print("# gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> "
          "fetch_x2 column")
    print("kernel,launches_per_chain,fetch_MB,fetch_x2_MB,write_MB")
    rows = []
    for k in set(fetch) | set(write):
        f = fetch.get(k, 0.0) * kib / chains
        w = write.get(k, 0.0) * kib / chains
        rows.append((f * 2 + w, k, calls.get(k, 0) / chains, f, w))
    tf = tw = 0.0
    for _, k, c, f, w in sorted(rows, reverse=True):
        if f * 2 + w < 0.05:
            continue
        print(f'"{k}",{c:.2f},{f:.1f},{2 * f:.1f},{w:.1f}')
        tf += f
        tw += w
    print(f"TOTAL,,{tf:.1f},{2 * tf:.1f},{tw:.1f}")


if __name__ == "__main__":
    main()
