set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3; mkdir -p $O
B="--streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d"
cd $R
for pad in 0 8192 16384 32768; do
  AMT_FORK=0 AMT_WS_ANYORDER=0 AMT_WS_LDS_PAD=$pad timeout -k 10 200 python3 bench.py $B > $O/pad_${pad}.json 2> $O/pad_${pad}.err || exit 1
done
echo "pads done"
cd /tmp && export TMPDIR=/tmp
AMT_FORK=0 AMT_WS_ANYORDER=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_inorder -o k -- python3 $R/bench.py $B > $O/ks_inorder.log 2>&1 || exit 1
AMT_FORK=0 AMT_WS_ANYORDER=0 AMT_WS_LDS_PAD=16384 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_pad16 -o k -- python3 $R/bench.py $B > $O/ks_pad16.log 2>&1 || exit 1
echo "stats done"
P="--streams 1 --batch 48 --steps 2 --warmup 1 --no-cpu --no-h2d"
AMT_FORK=0 AMT_WS_ANYORDER=0 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/pmc_sq -o sq --output-format csv -- python3 $R/bench.py $P > $O/pmc_sq.log 2>&1 || exit 1
echo "pmc done"
AMT_FORK=0 AMT_WS_ANYORDER=0 timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT -d $O/pmc_sq2 -o sq --output-format csv -- python3 $R/bench.py $P > $O/pmc_sq2.log 2>&1; echo "pmc2 rc=$?"
