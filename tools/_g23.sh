set -o pipefail
O=gpurun_out/r3; mkdir -p $O
CAMPAIGN_FIRST=700 CAMPAIGN_LAST=724 timeout -k 10 900 python3 tests/campaigns/parity_campaign.py > $O/campaign_700.log 2>&1; echo "campaign rc=$?"; tail -3 $O/campaign_700.log
timeout -k 10 600 python3 tests/campaigns/size_sweep.py > $O/size_sweep.log 2>&1; echo "size sweep rc=$?"; tail -2 $O/size_sweep.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $O/t_all.log
