"""Import-compatible stand-ins for the third-party modules LEMO's hot path imports.

    import lemo_amd.compat.smplx as smplx         # smplx.create / smplx.lbs.lbs / smplx.lbs.transform_mat
    import lemo_amd.compat.chamfer as chamfer     # importable, never called under the S2 / S3 configurations

``install()`` registers them under the reference's own import names so that ``import smplx`` /
``from smplx.lbs import lbs`` / ``import chamfer`` inside LEMO resolve here.
"""
import sys


def install(force: bool = False) -> None:
    """``sys.modules['smplx']``, ``['smplx.lbs']`` and ``['chamfer']`` -> this package (existing entries are kept
    unless ``force``)."""
    from . import chamfer, smplx
    from .smplx import lbs
    for name, mod in (('smplx', smplx), ('smplx.lbs', lbs), ('chamfer', chamfer)):
        if force or name not in sys.modules:
            sys.modules[name] = mod
