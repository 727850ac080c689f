set -o pipefail
mkdir -p gpurun_out/r3
CAMPAIGN_FIRST=40 CAMPAIGN_LAST=52 timeout -k 10 900 python3 tests/campaigns/parity_campaign.py > gpurun_out/r3/parity2.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/r3/parity2.log
timeout -k 10 400 python3 tests/campaigns/fuzz_tiny.py 200 > gpurun_out/r3/fuzz_tiny.log 2>&1; echo "tiny rc=$?"; tail -1 gpurun_out/r3/fuzz_tiny.log
timeout -k 10 400 python3 tests/campaigns/fuzz_dynamics.py 40 > gpurun_out/r3/fuzz_dyn.log 2>&1; echo "dyn rc=$?"; tail -1 gpurun_out/r3/fuzz_dyn.log
timeout -k 10 400 python3 tests/campaigns/soak_concurrent.py 20 > gpurun_out/r3/soak.log 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r3/soak.log
timeout -k 10 400 python3 tests/campaigns/size_sweep.py > gpurun_out/r3/sweep.log 2>&1; echo "sweep rc=$?"; tail -1 gpurun_out/r3/sweep.log
