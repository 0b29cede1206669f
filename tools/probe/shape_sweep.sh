#!/bin/bash
# Row-shape sweep of the fluid-only kernels (GPU box): same ny x nz = 512 x 512 planes (64 for the longest rows),
# row length varied -- isolates the effect of the workgroup shape (waves per row) from the domain size.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for d in 64x512x512 128x512x512 256x512x512 384x512x512 512x512x512 576x512x512 640x512x512 768x512x512 1024x512x256 1024x512x64 1536x256x256 2048x256x256; do
  B=576; [ ${d%%x*} -gt 1024 ] && B=256     # block: what the library picks by default for that row length
  timeout 300 python tools/perf_probe.py --dims $d --variants ${VARIANTS:-11} --blocks $B --reps ${REPS:-40} --modes even,odd,ab 2>&1 | tail -n +2 | sed "s/^/$d  /"
done
