mkdir -p gpurun_out
for a in 0 2; do echo "== cot_set_tuning(24, $a)"; COT_TUNING="24=$a" python scripts/bench_conv_abi.py --only g4 --modes 1 2>&1 | grep -E "g4|grouped"; done
