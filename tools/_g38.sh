set -o pipefail
mkdir -p gpurun_out/r3
for g in 512 256 128 64; do
AMT_FORK=0 AMT_HIST_GRID=$g timeout -k 10 300 python3 bench.py --streams 1 --batch 48 --steps 5 --warmup 2 --no-sublines --no-cpu --no-h2d > gpurun_out/r3/hg_$g.json 2> gpurun_out/r3/hg_$g.err; echo "hist grid $g rc=$?"
grep "stage ms" gpurun_out/r3/hg_$g.err | tail -1
done
