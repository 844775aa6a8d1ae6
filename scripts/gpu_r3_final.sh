#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -5 > $O/r3final_pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/r3final_pytest.log
timeout 300 python bench.py > $O/r3final_bench.json 2> $O/r3final_bench.err
