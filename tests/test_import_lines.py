"""The reference drivers' own import lines must resolve against this repository unchanged."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# run_demo.py:27-32 and train_material_params.py:28-33, verbatim (the ``warp`` import resolves to the four-name facade
# when mpmavatar_amd/compat is on the path; with NVIDIA Warp installed it would resolve to Warp and work just as well)
DRIVER_IMPORTS = '''
import warp as wp
from warp_mpm.mpm_data_structure import (
    MPMStateStruct,
    MPMModelStruct,
)
from warp_mpm.mpm_solver import MPMWARP
'''

CHECK = DRIVER_IMPORTS + '''
import mpmavatar_amd.warp_mpm as shim
assert MPMWARP is shim.MPMWARP and MPMStateStruct is shim.MPMStateStruct and MPMModelStruct is shim.MPMModelStruct
import torch
t = torch.zeros(4, 3)
assert wp.to_torch(t) is t          # run_demo.py:532: wp.to_torch(self.mpm_state.particle_x).clone()
assert hasattr(wp, "init") and hasattr(wp, "config")
for name in ("p2g2p", "set_parameters_dict", "set_E_nu_from_torch", "prepare_mu_lam", "add_mesh_collider", "add_particle_mover",
             "add_surface_collider", "add_bounding_box", "set_velocity_on_cuboid", "print_time_profile"):
    assert callable(getattr(MPMWARP, name)), name
for name in ("init", "from_torch", "reset_state", "continue_from_torch", "reset_density"):
    assert callable(getattr(MPMStateStruct, name)), name
print("ok")
'''


def test_reference_import_lines_resolve_unchanged(tmp_path):
    """Run from a directory laid out like the reference's root: a ``warp_mpm/`` directory WITHOUT ``__init__.py`` holding its
    own mpm_solver.py (a namespace package, first on sys.path as the script's cwd).  The regular package of this repository,
    later on the path, must win."""
    fake = tmp_path / "refroot"
    (fake / "warp_mpm").mkdir(parents=True)
    (fake / "warp_mpm" / "mpm_solver.py").write_text("raise ImportError('the reference Warp solver must not be imported')\n")
    (fake / "warp_mpm" / "mpm_data_structure.py").write_text("raise ImportError('the reference Warp structs must not be imported')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "mpmavatar_amd", "compat")]))
    r = subprocess.run([sys.executable, "-c", CHECK], cwd=str(fake), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_facade_from_the_package():
    env = dict(os.environ, PYTHONPATH=ROOT)
    code = "from warp_mpm import wp; import torch; t = torch.ones(2); assert wp.to_torch(t) is t; print('ok')"
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_hardware_queues_are_requested_by_the_fd_step_not_by_the_import():
    """importing the package must not change the queue behaviour of every HIP user of the host process.  The concurrent FD
    step asks for one hardware queue per stream itself (fd.request_hw_queues): sets GPU_MAX_HW_QUEUES when nothing has touched the
    device yet, keeps a value the user chose."""
    import subprocess, sys
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import mpmavatar_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'))"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip() == "None"
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); from mpmavatar_amd import fd; print(os.environ.get('GPU_MAX_HW_QUEUES'), "
            "fd.request_hw_queues(), os.environ['GPU_MAX_HW_QUEUES'])")
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split() == ["None", "True", "8"]
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '2'; from mpmavatar_amd import fd; print(fd.request_hw_queues(), os.environ['GPU_MAX_HW_QUEUES'])"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split() == ["True", "2"]
