#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${1:+-k "$1"} 2>&1 | tail -25
