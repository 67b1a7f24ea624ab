"""Drop-in for the reference's pybind11 extension of the same name (imported at
mdqe/models/ops/functions/ms_deform_attn_func.py:19; built there by mdqe/models/ops/setup.py from src/vision.cpp:13-16):
put this directory on PYTHONPATH and `import MultiScaleDeformableAttention as MSDA` resolves to the MI355X op
(Option A)."""
from mdqe_cvpr2023_amd.MultiScaleDeformableAttention import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
