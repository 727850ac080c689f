set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 120 ./tools/probes/anyorder_probe.bin > $O/anyorder.log 2>&1; echo "probe rc=$?"
B="--streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d"
AMT_FORK=0 AMT_WS_ANYORDER=0 timeout -k 10 300 python3 bench.py $B > $O/b48_inorder.json 2> $O/b48_inorder.err && echo "b48 inorder ok" &&
AMT_FORK=0 AMT_WS_ANYORDER=1 timeout -k 10 300 python3 bench.py $B > $O/b48_any.json 2> $O/b48_any.err && echo "b48 any ok" &&
AMT_WS_ANYORDER=0 timeout -k 10 300 python3 bench.py --no-cpu --no-h2d > $O/def_inorder.json 2> $O/def_inorder.err && echo "def inorder ok" &&
AMT_WS_ANYORDER=1 timeout -k 10 300 python3 bench.py --no-cpu --no-h2d > $O/def_any.json 2> $O/def_any.err && echo "def any ok" &&
AMT_WS_ANYORDER=1 timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d > $O/p48_any.json 2> $O/p48_any.err && echo "p48 any ok" &&
AMT_WS_ANYORDER=0 timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d > $O/p48_inorder.json 2> $O/p48_inorder.err && echo "p48 inorder ok" &&
AMT_WS_ANYORDER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_chain.py tests/test_gpu_ops.py -m gpu -x -q -k "watershed or chain or c3 or fused" > $O/t_any.log 2>&1; echo "tests rc=$?"; tail -3 $O/t_any.log
