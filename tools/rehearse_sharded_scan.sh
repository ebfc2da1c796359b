#!/bin/bash
# Rehearsal of the view-sharded pipeline on a ONE-GPU box: N ranks share cuda:0, gloo collectives
# (broadcast flavour of the all-gatherv: gloo has no CUDA send/recv).  The sharded runs must write the
# same model, byte for byte, as the single-process run.
set -e
T=$(mktemp -d)
python - "$T" <<'PY'
import sys
from pathlib import Path
sys.path.insert(0, "tests")
from scan_factory import make_scan
make_scan(Path(sys.argv[1]) / "scans", "plane", V=7, H=96, W=128, seed=3, floaters=0.03)
PY
S=$T/scans/plane
ARGS="--paths.recon-path $S/sparse/0 --paths.image-dir $S/images --moge.cache-dir $S/moge_cache --processing.downsample-density 2 --refiner.no-use-fp16 --refiner.no-adaptive-correspondences --filtering.vote-threshold 2 --refiner.verbose 0"
python scripts/test.py $ARGS --paths.output-model-dir $T/out1 > $T/log1.txt 2>&1
grep "Filtering removed\|number of dense" $T/log1.txt
for N in 2 3; do
  DD_DIST_BACKEND=gloo DD_ALLGATHERV=broadcast timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $((29700 + N)) scripts/test.py $ARGS --paths.output-model-dir $T/out$N > $T/log$N.txt 2>&1 \
      || { tail -30 $T/log$N.txt; exit 1; }
  grep "Sharding\|Filtering removed\|number of dense\|Sharded filter" $T/log$N.txt
  for f in cameras.bin images.bin points3D.bin; do cmp $T/out1/$f $T/out$N/$f; done
  echo "world $N: model identical to the single-GPU run (every rank wrote its own slice of points3D.bin)"
  # the same with the kept clouds gathered to rank 0 (xyz + rgba records over the wire), rank 0 writing alone
  DD_DIST_BACKEND=gloo DD_ALLGATHERV=broadcast timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $((29710 + N)) scripts/test.py $ARGS --processing.no-sharded-model-write \
      --paths.output-model-dir $T/outg$N > $T/logg$N.txt 2>&1 || { tail -30 $T/logg$N.txt; exit 1; }
  for f in cameras.bin images.bin points3D.bin; do cmp $T/out1/$f $T/outg$N/$f; done
  echo "world $N: gathered write identical too"
done
