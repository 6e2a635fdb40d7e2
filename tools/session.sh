#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
