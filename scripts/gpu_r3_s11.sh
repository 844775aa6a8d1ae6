#!/bin/bash
# round 3, session 11: new parity tests (real-geometry reference fixtures, N = 80 oracle comparison, per-stage forward-only bounds), smoke()
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 1200 python -m pytest tests/test_layers_gpu.py tests/test_fused_bn_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "real_stage or benchmark_batch or forward_only" > $O/r3s11_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r3s11_pytest.log
grep -E "passed|failed|^FAILED|^E  |rc=" $O/r3s11_pytest.log | cut -c1-400 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r3s11_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r3s11_smoke.log; grep -v amdgpu $O/r3s11_smoke.log | tail -5
