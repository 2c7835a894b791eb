#!/bin/bash
# several ranks sharing ONE GPU through the in-library loop (mock RCCL, peer-mapped + fused halo): does splitting the scene into
# concurrently running shards hide the per-kernel tails?
cd $GRAFT_REPO_ROOT
MOCK=$(python -c "import sys; sys.path.insert(0,'tests/mock_rccl'); from build import build; print(build())")
for n in 2 3 4; do
  MPMHIP_DIST_BACKEND=gloo OMP_NUM_THREADS=1 MPMHIP_DIST_TRANSPORT=rccl MPMHIP_RCCL_LIB=$MOCK timeout 600 python bench.py --gpus $n --steps 400 --warmup 40 --no-cpu-baseline --advance 0 --no-weak 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ranks on one GPU: $n', round(d['value']), round(d['ms_per_step']*1e3,1), 'us', d['exchange']['halo'], [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:4]])"
done
