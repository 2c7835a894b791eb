#!/bin/bash
# round 3, batch e: whole GPU suite with the tightened bounds; full-size parity record of the S3 garment (1000 substeps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
python tools/gpu/full_parity.py garment-120k-aniso 1000 > $O/full_parity_garment.log 2>&1; tail -14 $O/full_parity_garment.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
