"""Mirror of the reference's `panst3r.engine` exports that are on (or next to) the hot path:
`from panst3r_amd.engine import panoptic_inference_v2` replaces `from panst3r.engine import panoptic_inference_v2`
(engine/__init__.py, tools/demo_panst3r.py:41)."""
from .postprocess import panoptic_inference_v2, panoptic_inference_v1, panoptic_inference_qubo  # noqa: F401
from . import pointmaps  # noqa: F401,E402  (pointmap post-processing: postprocess / estimate_focal_knowing_depth / rigid_points_registration)
from .images import load_images  # noqa: F401,E402
