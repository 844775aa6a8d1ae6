mkdir -p gpurun_out
( timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/final2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final2_pytest.log )
( timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final2_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final2_smoke.log )
tail -n 3 gpurun_out/final2_pytest.log; tail gpurun_out/final2_smoke.log
