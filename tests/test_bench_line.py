"""CPU checks of what bench.py SAYS about itself (the headline line named fp32-input MFMA while the shipped
variant 9 multiplies on split-f16)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('lemo_bench', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize('variant', list(range(0, 12)))
def test_arithmetic_label_names_the_scheme_that_runs(variant):
    label = _bench().arithmetic_label(variant)
    if variant >= 4:
        assert 'split-f16' in label and 'fp16 pieces' in label and 'fp32-input MFMA' not in label
    elif variant == 3:
        assert 'bf16 pieces' in label
    else:
        assert 'fp32-input MFMA' in label


def test_default_variant_label_is_split_f16():
    b = _bench()
    from lemo_amd.priors import DEFAULT_CONV_VARIANT
    assert b.DEFAULT_CONV_VARIANT == DEFAULT_CONV_VARIANT
    assert 'split-f16' in b.arithmetic_label(DEFAULT_CONV_VARIANT)
