#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_runner.py -x -q -k "open_channel or other_density or poiseuille or channel" 2>&1 | tail -8
