cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03s_pytest_full.log 2>&1; grep -n "passed\|failed\|Error\|assert" gpurun_out/r03s_pytest_full.log | head -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
