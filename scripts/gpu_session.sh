#!/bin/bash
# one gpurun session: $1 = tag, rest = what to run (tests | bench | prof ...); writes gpurun_out/<tag>_*
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out; T=$1; shift
for what in "$@"; do
case $what in
  newtests) timeout 900 python -m pytest tests/test_side_stream_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider -x --timeout 300 2>&1 | tail -40 > $O/${T}_newtests.log ;;
  newtests_all) timeout 1200 python -m pytest tests/test_side_stream_gpu.py tests/test_dispatch_parity_gpu.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -80 > $O/${T}_newtests.log ;;
  tests) timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -40 > $O/${T}_pytest.log ;;
  smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/${T}_smoke.log ;;
  bench) timeout 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err ;;
esac
done
