#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for m in "none" "eager" "graph" "eager free"; do timeout 400 python scripts/r5_def_context.py $m 2>/dev/null | tail -1; done
