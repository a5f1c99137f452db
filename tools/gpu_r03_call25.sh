#!/bin/bash
# round-3 GPU call 25: wait / cache / LDS diagnostics of the final build's kernels (C1, C4) for the record
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/pmc_diag.sh r03final C1 2>&1 | cut -c1-1500
bash tools/pmc_diag.sh r03final C4 2>&1 | cut -c1-1500
