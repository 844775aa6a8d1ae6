#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
COT_NO_PROBE_CACHE=1 timeout 400 python bench.py --kernels new --no-cpu-baseline --graph on > $O/r3s39_on.json 2> $O/r3s39_on.err
timeout 400 python bench.py > $O/r3s39_default.json 2> $O/r3s39_default.err
