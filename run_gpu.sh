#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
tail -6 gpurun_out/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log | cut -c1-1500
