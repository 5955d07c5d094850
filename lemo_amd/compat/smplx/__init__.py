"""``smplx`` as LEMO uses it: ``create`` (opt_amass_temp.py:73-87, temp_prox/main_slide.py:160-179) and the ``lbs``
sub-module (human_body_prior/body_model/body_model.py:29, temp_prox/camera.py:27)."""
from ...body_model import SMPLX, ModelOutput, create          # noqa: F401
from . import lbs                                               # noqa: F401
