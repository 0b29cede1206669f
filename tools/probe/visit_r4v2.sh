export GPU_TAG=r4v2
O=$PWD/gpurun_out/r4v2; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for cfg in "AA 4 x" "AA 1 x" "AB 4 x" "AA 1 z"; do
  set -- $cfg
  tag=$1_K$2_$3
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -o t -- \
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $R/tools/probe/xslab_run.py $1 $2 200 $3 2>&1 | grep xslab_run )
  echo "== $tag"; python tools/probe/trace_gaps.py $O/trace_$tag 100 | tee $O/gaps_$tag.txt
  # keep the raw trace small: first 3000 lines
  for f in $(find $O/trace_$tag -name "*kernel_trace.csv"); do head -3000 $f > $O/kernel_trace_$tag.head.csv; done
  rm -rf $O/trace_$tag
done
