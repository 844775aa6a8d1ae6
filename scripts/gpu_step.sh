#!/bin/bash
TAG=${1:-step}
mkdir -p gpurun_out
timeout 420 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --kernels new > gpurun_out/${TAG}_new.json 2> gpurun_out/${TAG}_new.err; cut -c1-220 gpurun_out/${TAG}_new.json
