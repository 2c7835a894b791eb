cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for w in 0 1; do
  for scene in sheet-500k garment-120k-aniso; do
    MPMHIP_W6=$w timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('w6=$w $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])"
  done
done
done
