"""rocprofv3 --kernel-trace --stats output (*_kernel_stats.csv) -> the compact table committed under profiles/.

    python tools/kernel_stats_summary.py gpurun_out/prof_c3/c3_kernel_stats.csv "<bench args>" > profiles/r01_kernel_stats.csv
"""
import csv
import sys


def main():
    path, args = sys.argv[1], sys.argv[2]
    print(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {args}")
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    note = sys.argv[3] if len(sys.argv) > 3 else "# one launch = one stage kernel over the batch of 2048x2048 planes named in the command"
    if any("ws_flood" in r["Name"] for r in rows):
        note += ("; rocprofv3's kernel trace serialises the launches, so ws_flood_persist_kernel / ws_flood_lds_kernel / "
                 "ws_flood_edt_kernel (any-order launches that overlap in an unprofiled run) appear one after the other")
    print(note)
    print("kernel,calls,avg_us,total_ms,pct")
    if True:
        for r in rows:
            name = r["Name"].replace('"', "'")
            print(f'"{name}",{r["Calls"]},{float(r["AverageNs"]) / 1e3:.1f},{float(r["TotalDurationNs"]) / 1e6:.2f},'
                  f'{float(r["Percentage"]):.2f}')


if __name__ == "__main__":
    main()
