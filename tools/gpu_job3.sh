#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$1" 2>&1 | tail -25
