#!/bin/bash
# usage: r05_tests.sh TAG "pytest args"   -- a GPU pytest session whose log lands in gpurun_out/
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
TAG=$1; shift
timeout 1200 python -m pytest "$@" -m gpu -q --timeout 300 -rfE -p no:cacheprovider --tb=short > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.log
tail -60 $O/${TAG}_pytest.log | cut -c1-300
