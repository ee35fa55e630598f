#!/bin/bash
# Run ON THE GPU BOX: host-side profile of the training bench step (where the Python time goes).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m cProfile -o /tmp/train.prof bench.py --train --steps 10 --warmup 3 > /tmp/train.log 2>&1
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/train.prof')
p.sort_stats('tottime').print_stats(28)
p.sort_stats('cumulative').print_stats('video-k-net_amd|bench.py', 40)
PY
