#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
( time timeout 600 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/r3s41_cotnext101_auto.json 2> $O/r3s41_cotnext101_auto.err ) 2> $O/r3s41_cotnext101.time
( time timeout 600 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/r3s41_secotnetd152_auto.json 2> $O/r3s41_secotnetd152_auto.err ) 2> $O/r3s41_secotnetd152.time
