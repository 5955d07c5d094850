"""Importable stand-in for the ``chamfer`` CUDA extension (temp_prox/dist_chamfer.py:27,43).

LEMO's fitting configurations S2 / S3 set the Chamfer term's weight to 0 and never reach it; the
module only has to import.  Calling it is an error, not a silent CPU path.
"""


def forward(*args, **kwargs):
    raise NotImplementedError('chamfer.forward: the Chamfer term is disabled in LEMO\'s S2/S3 configurations and is '
                              'not part of the MI355X hot path')


def backward(*args, **kwargs):
    raise NotImplementedError('chamfer.backward: see chamfer.forward')
