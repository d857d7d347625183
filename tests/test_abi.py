"""The C-ABI library builds, loads on a CPU-only host and exports every symbol include/rmem.h declares
(no compute calls here: there is no GPU in this tier)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def lib():
    from rmem_ocu_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rmem.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rmem_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from rmem_ocu_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/rmem.h but not exported by librmem_hip.so'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in rmem_ocu_amd/_lib.py'
    assert set(_lib.SIGNATURES) <= set(names), set(_lib.SIGNATURES) - set(names)
    # the IEEE-half flavour: one twin per entry point with 16-bit operands, none for the element-type-agnostic ones
    twins = {n[:-4] for n in names if n.endswith('_f16')}
    assert twins == set(_lib.F16_TWINS)
    assert all(_lib.SIGNATURES[n + '_f16'] == _lib.SIGNATURES[n] for n in twins)


def test_version_and_error_string(lib):
    from rmem_ocu_amd import _lib
    assert lib.rmem_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.rmem_last_error_string(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    """Bad arguments are rejected on the host before anything is launched."""
    from rmem_ocu_amd._lib import ConvDesc
    d = ConvDesc(8, 8, 12, 8, 8, 16, 1, 1, 1, 0, 16, 16, 16, 0, 0, 0)      # Cin not a multiple of 8
    rc = lib.rmem_conv2d_nhwc(ctypes.byref(d), 16, 16, None, None, 16, None, None, None)
    assert rc != 0 and b'multiple of 8' in lib.rmem_last_error_string()
    rc = lib.rmem_mem_read_attn(16, 256, 16, 16, 0, 256, None, 1, 0, None, None, 10, 8, 16, 256, None, 0, 16, None)
    assert rc != 0 and b'lk_single' in lib.rmem_last_error_string()
    assert lib.rmem_attn_workspace_bytes(1674, 8, 8) == 8 * 8 * 1674 * 36 * 4


def test_no_cpu_fallback():
    """The product path refuses host tensors instead of silently computing on the CPU."""
    import torch
    from rmem_ocu_amd import ops
    from rmem_ocu_amd._lib import RmemError
    x = torch.zeros(16, 8, dtype=torch.bfloat16)
    with pytest.raises(RmemError):
        ops.add16(x, x, x, 128)
    from rmem_ocu_amd import build_vos_model, get_config
    model = build_vos_model('aot', get_config())
    with pytest.raises(RuntimeError):
        model.packed()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'rmem_ocu_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f'{f} imports the oracle'
                assert 'ref_cpu' not in src, f'{f} refers to the oracle module'
