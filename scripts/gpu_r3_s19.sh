#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/wgrad_stamps.py > $O/r3s19_stamps_pf.log 2>&1
timeout 300 python scripts/wgrad_stamps.py --tune 25=2 > $O/r3s19_stamps_nopf.log 2>&1
