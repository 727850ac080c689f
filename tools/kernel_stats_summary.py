"""rocprofv3 --kernel-trace --stats output (*_kernel_stats.csv) -> the compact table committed under profiles/.

    python tools/kernel_stats_summary.py gpurun_out/prof_c3/c3_kernel_stats.csv "<bench args>" > profiles/r01_kernel_stats.csv
"""
import csv
import sys


def main():
    path, args = sys.argv[1], sys.argv[2]
    print(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {args}")
    print("# one launch = one stage kernel over a batch of 32 FOVs (32 planes of 2048x2048); the three ws_flood_lds "
          "classes run concurrently")
    print("kernel,calls,avg_us,total_ms,pct")
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Name"].replace('"', "'")
            print(f'"{name}",{r["Calls"]},{float(r["AverageNs"]) / 1e3:.1f},{float(r["TotalDurationNs"]) / 1e6:.2f},'
                  f'{float(r["Percentage"]):.2f}')


if __name__ == "__main__":
    main()
