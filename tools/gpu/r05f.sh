#!/bin/bash
mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests/test_gpu_stress_ahead.py tests/test_gpu_parity.py tests/test_gpu_ref_golden.py -m gpu -x -q --durations=5 > gpurun_out/r05f/parity.txt 2>&1; echo "rc=$?" >> gpurun_out/r05f/parity.txt
tail -12 gpurun_out/r05f/parity.txt
python tools/gpu/ab5.py --libs default,saw4,saw5,default@MPMHIP_STRESS_AHEAD=0 --scenes sheet-500k,garment-120k-aniso --reps 2 --advance 2000 --out gpurun_out/r05f/ab.json > gpurun_out/r05f/ab.txt 2>&1
cat gpurun_out/r05f/ab.txt
torchrun_log=gpurun_out/r05f/dist8.txt
MPMHIP_TEST_REBIN=8 MPMHIP_DIST_HALO=peer MPMHIP_VERBOSE=1 MPMHIP_DIST_TRANSPORT=rccl MPMHIP_RCCL_LIB=$PWD/tests/mock_rccl/librccl_mock.so OMP_NUM_THREADS=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29517 tests/dist_worker.py gpu widesheet8 40 > $torchrun_log 2>&1; echo "rc=$?" >> $torchrun_log
grep -v "Gloo\|^Particles\|^Total" $torchrun_log | grep -i "error\|Traceback\|rank\|File\|rc=" | head -40
