#!/bin/bash
# round-3 GPU call 18: full GPU suite on the final tree (facade in worldpoint mode), smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-400
