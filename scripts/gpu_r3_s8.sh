#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python scripts/wgrad_stamps.py > $O/r3s8_stamps.log 2>&1; grep -v amdgpu $O/r3s8_stamps.log
