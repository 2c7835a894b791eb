#!/usr/bin/env python
"""Per-particle deviation of the spinning jelly cube from the reference's sequence fixture, per library variant:
    python tools/gpu/pp_jelly.py default nofma nofma_p2g@MPMHIP_G2P2G=0 ...     (name[@K=V;K=V]: lib/variants/libmpmhip_<name>.so + environment)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import refgolden as rg
from mpmavatar_amd import harness
z = rg.load('ref_seq_cube_jelly')
out = {}
for mode in ('fast',):
    sim = harness.build_solver(rg.scene_from_npz(z), 'cuda:0', mode=mode)
    for cp in z['checkpoints']:
        harness.run(sim, int(cp) - sim.steps_done, fused=True)
    x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
    out[mode] = [rg.rel_pp(x, z[f's{cp}_particle_x']), rg.rel_pp(v, z[f's{cp}_particle_v']), rg.rel(v, z[f's{cp}_particle_v'])]
print('RESULT ' + json.dumps(out))
''' % (ROOT, os.path.join(ROOT, "tests"))
for spec in sys.argv[1:]:
    name, _, envs = spec.partition("@")
    env = dict(os.environ)
    if name != "default":
        env["MPMHIP_LIB"] = os.path.join(ROOT, "mpmavatar_amd", "lib", "variants", f"libmpmhip_{name}.so")
    for kv in filter(None, envs.split(";")):
        k, _, v = kv.partition("="); env[k] = v
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(spec, line[-1][7:] if line else "FAILED " + r.stderr[-400:], flush=True)
