#!/bin/bash
# full GPU parity suite, compact failure report
cd "${GRAFT_REPO_ROOT:-/root/repo}"; timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "^E    \|^    " | tail -${TAILN:-25}
