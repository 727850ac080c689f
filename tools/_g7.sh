set -o pipefail
AMT_API_CPROFILE=1 timeout -k 10 200 python3 tools/api_profile.py 1 2 2>&1 | tail -45
AMT_API_SAMPLE=1 timeout -k 10 200 python3 tools/api_profile.py 8 2 2>&1 | tail -30
timeout -k 10 300 python3 tools/deliver_probe.py 48 2>&1 | tail -8
timeout -k 10 300 python3 tools/deliver_probe.py 192 2>&1 | tail -8
