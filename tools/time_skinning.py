"""k_skin on a 4.2-million-vertex skinned mesh (the size at which it is bandwidth-bound): ten re-posings; run under
tools/kstats_cmd.sh for the kernel's duration. 160 algorithmic bytes per vertex (96 read + 64 written)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

r = Renderer(load_hip_capi(), 64, 64, default_settings(msaa=1))
geo = scenes.cylinder(0.3, 4.0, 2047, 2047).transformed(scenes.translation(0, 2.0, 0))
joints, weights = scenes.bone_chain_weights(geo.pos[:, 1].astype(np.float64), 32, 4.0)
mesh = r.add_mesh([scenes.upload_skinned_primitive(r, geo, joints, weights, scenes.make_material())], is_static=False)
rng = np.random.default_rng(5)
for k in range(10):
    pal = np.stack([(scenes.translation(*rng.normal(0, 0.1, 3)) @ scenes.rotation_y(rng.normal(0, 0.3))).T for _ in range(32)]).astype(np.float32)
    r.set_mesh_joints(mesh, pal)
    r.api.read_primitive_positions(r._ctx, mesh, 0, None, None)   # count only: no copy
    r.render_frame(exposure_mode=2, exposure=1.0)
print("vertices", geo.pos.shape[0], "algorithmic MB per launch", geo.pos.shape[0] * 160 / 1e6)
