#!/bin/bash
# one gpurun session: $1 = tag, rest = what to run (tests | bench | prof ...); writes gpurun_out/<tag>_*
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out; T=$1; shift
for what in "$@"; do
case $what in
  newtests) timeout 900 python -m pytest tests/test_side_stream_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider -x --timeout 300 2>&1 | tail -40 > $O/${T}_newtests.log ;;
  newtests_all) timeout 1200 python -m pytest tests/test_side_stream_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -80 > $O/${T}_newtests.log ;;
  tests) timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -40 > $O/${T}_pytest.log ;;
  smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/${T}_smoke.log ;;
  aggab1) timeout 600 python scripts/bench_agg_abi.py --dtypes bf16 --shapes 0 --rounds 3 --variants lds,d2_jp4_nw4_s1,d2_jp4_nw4_s0,d2_jp2_nw4_s1,d2_jp8_nw4_s1,d2_jp4_nw2_s1,d2_jp4_nw8_s1,d2_jp2_nw2_s1,d2_jp8_nw8_s1,d2_jp2_nw8_s1 > $O/${T}_aggab1.log 2>&1
          timeout 300 python scripts/bench_agg_abi.py --dtypes bf16 --shapes 1,2 --rounds 3 --variants lds,d2_jp4_nw4_s1,d2_jp4_nw4_s0,d2_jp2_nw4_s1,d2_jp8_nw4_s1,d2_jp4_nw2_s1,d2_jp4_nw8_s1,d2_jp4_nw4_s1_xcd >> $O/${T}_aggab1.log 2>&1 ;;
  aggab2) timeout 600 python scripts/bench_agg_abi.py --dtypes bf16 --shapes 0,1,2 --rounds 3 --variants lds,dot2,d2_jp2_nw4_s1,d2_jp2_nw7_s1,d2_jp4_nw7_s1,d2_jp2_nw7_s0,d2_jp2_nw7_s1_xcd,d2_jp2_nw4_s1_xcd > $O/${T}_aggab2.log 2>&1 ;;
  aggab3) timeout 600 python scripts/bench_agg_abi.py --dtypes bf16 --shapes 0,1,2 --rounds 3 --variants lds,d2_jp2_nw4_s1,d2_jp8_nw4_s1,d2_jp8_nw4_s0,d2_jp8_nw2_s1,d2_jp8_nw4_s1_xcd,d2_jp4_nw4_s1 > $O/${T}_aggab3.log 2>&1 ;;
  aggab4) timeout 600 python scripts/bench_agg_abi.py --dtypes bf16 --shapes 0,1,2 --rounds 3 --variants lds,d2_jp2_nw4_s1_sb,d2_jp2_nw4_s1,d2_jp4_nw4_s1,d2_jp2_nw2_s1,d2_jp4_nw2_s1,d2_jp2_nw4_s0,d2_jp2_nw4_s1_xcd,d2_jp2_nw7_s1 > $O/${T}_aggab4.log 2>&1 ;;
  coxt) timeout 900 python -m pytest tests/test_fused_layer_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -25 > $O/${T}_coxt_tests.log
        COT_KERNEL_SUMMARY=$O/${T}_cotnext101_kernels.json timeout 400 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/${T}_bench_cotnext101.json 2> $O/${T}_bench_cotnext101.err
        timeout 400 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/${T}_bench_cotnext101_b.json 2>> $O/${T}_bench_cotnext101.err
        COT_FUSED_LAYER=0 timeout 400 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --conv1x1 hip --conv3x3 hip --gn9 > $O/${T}_bench_cotnext101_nodeperop.json 2>> $O/${T}_bench_cotnext101.err
        COT_KERNEL_SUMMARY=$O/${T}_secotnetd_kernels.json timeout 400 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/${T}_bench_secotnetd.json 2> $O/${T}_bench_secotnetd.err ;;
  diagcoxt) timeout 300 python scripts/diag_coxt_fixture.py > $O/${T}_diag_coxt.log 2>&1 ;;
  bench_sec) timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err ;;
  secotnetd) timeout 400 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/${T}_bench_secotnetd.json 2> $O/${T}_bench_secotnetd.err
             timeout 300 python -m pytest tests/test_se_gate_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "se_cotnetd or hybrid or model" 2>&1 | tail -8 > $O/${T}_secotnetd_tests.log ;;
  tunes) for tn in "35=4" "35=5" "35=4" "35=5"; do timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 --tune $tn 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$tn', l['value'], l['ms_per_step'], r['kernel'], r['avg_us'], r['frac'], [ (k['shape'],k['avg_us']) for k in r['kernels'] if 'bwd' in k['kernel']])" >> $O/${T}_tunes.log 2>&1; done ;;
  layout) timeout 600 python scripts/bench_layout_study.py --json $O/${T}_layout.json > $O/${T}_layout.log 2>&1 ;;
  grouped) timeout 600 python -m pytest tests/test_conv_general_gpu.py tests/test_fused_layer_gpu.py tests/test_conv1x1_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -8 > $O/${T}_grouped_tests.log
        COT_KERNEL_SUMMARY=$O/${T}_cotnext101_kernels.json timeout 400 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/${T}_bench_cotnext101_prof.json 2> $O/${T}_bench_cotnext101.err
        timeout 400 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/${T}_bench_cotnext101.json 2>> $O/${T}_bench_cotnext101.err
        timeout 400 python bench.py --model cotnext101_2x48d --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --tune 36=0 > $O/${T}_bench_cotnext101_general.json 2>> $O/${T}_bench_cotnext101.err ;;
  sec3x3) timeout 600 python -m pytest tests/test_conv3x3g_gpu.py tests/test_dispatch_parity_gpu.py tests/test_se_gate_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "conv3x3 or Conv3x3 or matches or se_" 2>&1 | tail -6 > $O/${T}_sec3x3_tests.log
        COT_KERNEL_SUMMARY=$O/${T}_secotnetd_kernels.json timeout 400 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/${T}_bench_secotnetd_prof.json 2> $O/${T}_bench_secotnetd.err
        timeout 400 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/${T}_bench_secotnetd.json 2>> $O/${T}_bench_secotnetd.err ;;
  sablock) timeout 600 python -m pytest tests/test_fused_layer_gpu.py tests/test_se_gate_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -8 > $O/${T}_sablock_tests.log
        COT_KERNEL_SUMMARY=$O/${T}_secotnetd_kernels.json timeout 400 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/${T}_bench_secotnetd_prof.json 2> $O/${T}_bench_secotnetd.err
        timeout 400 python bench.py --model se_cotnetd_152_L --img 320 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/${T}_bench_secotnetd.json 2>> $O/${T}_bench_secotnetd.err ;;
  gnfuse) timeout 600 python -m pytest tests/test_gn_fusion_gpu.py tests/test_fused_layer_gpu.py tests/test_group_norm9_gpu.py tests/test_conv1x1_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -8 > $O/${T}_gnfuse_tests.log
        for gf in 1 0 1 0; do COT_GN_FUSED=$gf timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('COT_GN_FUSED=$gf', l['value'], l['ms_per_step'])" >> $O/${T}_gnfuse_ab.log 2>&1; done ;;
  libab) for lp in cotnet_amd/lib/libcotnet_hip_2786414.so "" cotnet_amd/lib/libcotnet_hip_2786414.so ""; do COT_GN_FUSED=0 COT_LIB_PATH=$lp timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('lib=${lp:-HEAD}', l['value'], l['ms_per_step'])" >> $O/${T}_libab.log 2>&1; done
        COT_KERNEL_SUMMARY=$O/${T}_head_kernels.json timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
        COT_GN_FUSED=0 COT_LIB_PATH=cotnet_amd/lib/libcotnet_hip_2786414.so COT_KERNEL_SUMMARY=$O/${T}_old_kernels.json timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1 ;;
  aggparity) timeout 600 python -m pytest tests/test_dispatch_parity_gpu.py tests/test_agg_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "agg or Agg or oracle or n80 or N80" 2>&1 | tail -30 > $O/${T}_aggparity.log ;;
  graph) timeout 200 python scripts/try_graph_step.py 10 cotnext101_2x48d 64 224 > $O/${T}_graph_cotnext.log 2>&1; tail -3 $O/${T}_graph_cotnext.log | cut -c1-300
         timeout 300 python scripts/try_graph_step.py 10 se_cotnetd_152_L 64 320 > $O/${T}_graph_secot.log 2>&1; tail -3 $O/${T}_graph_secot.log | cut -c1-300 ;;
  roofobj) timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print(l['value'], {k: r[k] for k in ('kernel','shape','achieved','frac','traffic','avg_us')}); [print(k) for k in r['kernels']]" ;;
  r4v) timeout 600 python -m pytest tests/test_pool_gpu.py tests/test_gn_fusion_gpu.py tests/test_fused_layer_gpu.py tests/test_agg_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -8 > $O/${T}_tests.log; tail -3 $O/${T}_tests.log
       bash scripts/gpu_trace_new.sh ${T}_secot --model se_cotnetd_152_L --img 320 --batch 64 > $O/${T}_trace_sh.log 2>&1; tail -2 $O/${T}_trace_sh.log | cut -c1-200 ;;
  r4w) timeout 600 python -m pytest tests/test_pool_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -8 > $O/${T}_tests.log; tail -3 $O/${T}_tests.log
       timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 --model se_cotnetd_152_L --img 320 --batch 64 2>/dev/null | cut -c1-200
       bash scripts/gpu_trace_new.sh ${T}_cotnext --model cotnext101_2x48d --batch 64 > $O/${T}_trace_sh.log 2>&1
       bash scripts/gpu_trace_new.sh ${T}_fp32 --dtype fp32 --batch 80 >> $O/${T}_trace_sh.log 2>&1 ;;
  ring) for r in 3 5 3 5; do echo "== ring $r"; timeout 200 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "g4" --tune 38=$r 2>/dev/null | grep "g4"; done > $O/${T}_ring.log 2>&1; cat $O/${T}_ring.log | cut -c1-120
        for r in 3 5; do COT_TUNING=38=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('ring $r', l['value'], l['ms_per_step'])"; done ;;
  res3) timeout 600 python -m pytest tests/test_conv3x3g_gpu.py tests/test_dispatch_parity_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "conv3x3 or layer or block" 2>&1 | tail -6 > $O/${T}_tests.log; tail -3 $O/${T}_tests.log
        for r in 0 1 0 1; do echo "== res $r"; timeout 200 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "g4" --tune 39=$r 2>/dev/null | grep "g4"; done > $O/${T}_res3.log 2>&1; cat $O/${T}_res3.log | cut -c1-120
        for r in 0 1 0 1; do COT_TUNING=39=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('res $r', l['value'], l['ms_per_step'])"; done
        for r in 0 1; do COT_TUNING=39=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 --model se_cotnetd_152_L --img 320 --batch 64 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('secot res $r', l['value'], l['ms_per_step'])"; done
        for r in 0 1; do COT_TUNING=39=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 --model cotnext101_2x48d --batch 64 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cotnext res $r', l['value'], l['ms_per_step'])"; done ;;
  parity2) timeout 900 python -m pytest tests/test_dispatch_parity_gpu.py tests/test_fused_bn_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -x 2>&1 | tail -15 > $O/${T}_parity2.log; tail -6 $O/${T}_parity2.log | cut -c1-300 ;;
  graphbench) for g in "" "--graph" "" "--graph"; do timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 $g 2> $O/${T}_graphbench.err | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('graph=[$g]', l['value'], 'images/s', l['ms_per_step'], 'ms/step; host issue per step', l['host_issue_ms_per_step'], 'ms (upper bound), one step into an idle queue', l['host_issue_ms_one_step_idle_queue'], 'ms; loss', l['final_loss'])" || tail -5 $O/${T}_graphbench.err; done
              for g in "" "--graph"; do timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 --model se_cotnetd_152_L --img 320 --batch 64 $g 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('se_cotnetd_152_L 320 B64 graph=[$g]', l['value'], 'images/s', l['ms_per_step'], 'ms/step; one step into an idle queue', l['host_issue_ms_one_step_idle_queue'], 'ms')"; done
              for g in "" "--graph"; do timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 --model cotnext101_2x48d --batch 64 $g 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cotnext101_2x48d B64 graph=[$g]', l['value'], 'images/s', l['ms_per_step'], 'ms/step; one step into an idle queue', l['host_issue_ms_one_step_idle_queue'], 'ms')"; done ;;
  bn7) timeout 600 python -m pytest tests/test_fused_bn_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "bn or 49" 2>&1 | tail -4
       for r in 0 1 0 1; do COT_TUNING=40=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); rows=[r for r in l['roofline']['conv_bn_calls'] if r['op'].startswith('bn') and 'HW49' in r['shape']]
print('key40=$r', l['value'], l['ms_per_step'], [(r['op'], r['shape'], r['ms_per_step'], r['frac_hbm']) for r in rows])"; done ;;
  stem) timeout 600 python -m pytest tests/test_stem_gpu.py tests/test_layers_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3
        for r in 0 1 0 1; do COT_TUNING=41=$r COT_KERNEL_SUMMARY=$O/${T}_k$r.json timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); k=json.load(open('$O/${T}_k$r.json'))['kernels']
print('key41=$r', l['value'], l['ms_per_step'], [(r['kernel'], r['avg_us']) for r in k if 'stem' in r['kernel']])"; done ;;
  mask) timeout 600 python -m pytest tests/test_fused_layer_gpu.py tests/test_fused_bn_gpu.py tests/test_side_stream_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3
        for r in 0 1 0 1; do COT_BN_RELU_MASK=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); f=l['roofline']['conv_bn_families']
print('relu_mask=$r', l['value'], l['ms_per_step'], 'bn_bwd', f['bn_bwd'], 'bn_fwd', f['bn_fwd'], 'loss', l['final_loss'])"; done ;;
  rtaps) timeout 600 python -m pytest tests/test_conv3x3g_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "conv3x3" 2>&1 | tail -3
         timeout 200 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "g4" 2>/dev/null | grep "g4"
         for r in 1 2; do COT_KERNEL_SUMMARY=$O/${T}_k.json timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read()); k=json.load(open('$O/${T}_k.json'))['kernels']
print('run $r', l['value'], l['ms_per_step'], [(r['kernel'], r['avg_us'], r['ms_per_step']) for r in k if 'reduce_taps' in r['kernel']])"; done ;;
  wsingle) timeout 600 python -m pytest tests/test_conv3x3g_gpu.py tests/test_dispatch_parity_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "conv3x3 or layer or block" 2>&1 | tail -2
           for r in 0 1 0 1; do echo "== key42 $r"; timeout 200 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "C256 g4" --tune 42=$r 2>/dev/null | grep "g4"; done
           for r in 0 1 0 1; do COT_TUNING=42=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('key42=$r', l['value'], l['ms_per_step'])"; done
           for r in 0 1; do COT_TUNING=42=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 --model cotnext101_2x48d --batch 64 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cotnext key42=$r', l['value'], l['ms_per_step'])"; done ;;
  ns3) timeout 600 python -m pytest tests/test_conv1x1_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "conv1x1" 2>&1 | tail -2
       for r in 0 1 0 1; do echo "== key43 $r"; timeout 200 python scripts/bench_conv_abi.py --iters 20 --only "s3" --tune 43=$r 2>/dev/null | grep "s3 "; done
       for r in 0 1 0 1; do COT_TUNING=43=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('key43=$r', l['value'], l['ms_per_step'])"; done ;;
  cols) timeout 600 python -m pytest tests/test_conv3x3g_gpu.py tests/test_dispatch_parity_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 -k "conv3x3 or layer or block" 2>&1 | tail -2
        for r in 0 1 0 1; do echo "== key44 $r"; timeout 200 python scripts/bench_conv_abi.py --iters 20 --modes 1 --only "C128 g4" --tune 44=$r 2>/dev/null | grep "g4"; done
        for r in 0 1 0 1; do COT_TUNING=44=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('key44=$r', l['value'], l['ms_per_step'])"; done
        for r in 0 1; do COT_TUNING=44=$r timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 --model cotnext101_2x48d --batch 64 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cotnext key44=$r', l['value'], l['ms_per_step'])"; done ;;
  bench) timeout 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err ;;
esac
done
