#!/bin/bash
# what an on-device migration event costs at 512,000 traditional particles, two ranks over the RCCL stand-in on the one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIB=$(python -c "import sys; sys.path.insert(0,'tests/mock_rccl'); from build import build; print(build())")
MPMHIP_VERBOSE=${MPMHIP_VERBOSE:-} MPMHIP_TEST_TRAD_MIG=0 MPMHIP_TEST_RUN_CHUNK=50 MPMHIP_DIST_TRANSPORT=rccl MPMHIP_RCCL_LIB=$LIB MPMHIP_TEST_REBIN=0 OMP_NUM_THREADS=1 HSA_ENABLE_IPC_MODE_LEGACY=0 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29517 tests/dist_worker.py gpu shear512k 200 2>&1 | grep -E "dist\[shear512k\]|migration event|Error|error" | cut -c1-300
