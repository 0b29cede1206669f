#!/bin/bash
# How stable is the placed-array rate from process to process, and how does it depend on the number of chunks / the
# span?  (tools/perf_probe.py allocates placed arrays like the product.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for d in ${DIMS:-256x256x256 128x512x512}; do
  for cfg in "16 72" "32 72" "16 144" "64 144" "8 36"; do
    set -- $cfg
    for rep in 1 2 3 4; do
      SLF_PLACEMENT_PARTS=$1 SLF_PLACEMENT_SPAN_GIB=$2 timeout 200 python tools/perf_probe.py --dims $d --variants 11 --blocks 576 --reps 60 --modes even,odd,ab 2>&1 | tail -1 | sed "s/^/$d parts $1 span $2  /" | cut -c1-170
    done
  done
done
