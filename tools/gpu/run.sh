#!/bin/bash
# tools/gpu/run.sh <tag> <what...> -- the one parametrised GPU-box runner (replaces the per-round r0*.sh batch scripts):
#   gate            full `pytest -m gpu -x -q` in the driver's order            -> gpurun_out/<tag>_pytest_gpu.txt
#   tests <expr>    `pytest -m gpu -x -q -k <expr>`                             -> gpurun_out/<tag>_pytest_k.txt
#   smoke           __graft_entry__.smoke()                                     -> gpurun_out/<tag>_smoke.txt
#   bench [args]    python bench.py [args] (default: no flags)                  -> gpurun_out/<tag>_bench.json / .log
#   prof            rocprofv3 --kernel-trace --stats of `bench.py --steps 200 --no-cpu-baseline --advance 0` -> gpurun_out/<tag>_prof/
# Several can be chained with `--`:  run.sh r06a tests "full_size or demo_250" -- bench -- prof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
while [ $# -gt 0 ]; do
  what=$1; shift
  args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  case $what in
    gate)  timeout 3000 python -m pytest tests/ -x -q -m gpu -s 2>&1 | tail -400 > gpurun_out/${tag}_pytest_gpu.txt; tail -5 gpurun_out/${tag}_pytest_gpu.txt ;;
    tests) timeout 3000 python -m pytest tests/ -x -q -m gpu -s -k "${args[0]}" 2>&1 | tail -300 > gpurun_out/${tag}_pytest_k.txt; tail -15 gpurun_out/${tag}_pytest_k.txt ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1; tail -3 gpurun_out/${tag}_smoke.txt ;;
    bench) timeout 1200 python bench.py "${args[@]}" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log; tail -c 1500 gpurun_out/${tag}_bench.json; grep "bench +" gpurun_out/${tag}_bench.log | tail -12 ;;
    prof)  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/${tag}_prof -o p -- python $OLDPWD/bench.py --steps 200 --no-cpu-baseline --advance 0 "${args[@]}" > $OLDPWD/gpurun_out/${tag}_prof.json 2> $OLDPWD/gpurun_out/${tag}_prof.log)
           f=$(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" ;;
    *) echo "unknown: $what"; exit 2 ;;
  esac
done
