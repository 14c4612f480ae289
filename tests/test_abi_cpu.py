"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/vince_hip.h declares; argument
validation paths return error codes without touching a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from vince_amd import build
    build.build(verbose=False)
    from vince_amd import _lib
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    from vince_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vince_hip.h")).read()
    declared = set(re.findall(r"\b(vince_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    for name in declared:
        assert hasattr(L, name)
    # one number in three places: the header's VINCE_ABI_VERSION, what the built library returns, what the bindings expect
    version = int(re.search(r"#define\s+VINCE_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert L.vince_abi_version() == version == _lib.ABI_VERSION


def test_argument_validation_returns_codes(L):
    from vince_amd._lib import ConvDesc, InfoNCEDesc, TrunkCfg
    d = ConvDesc(N=1, Hi=4, Wi=4, Ci=6, Ho=4, Wo=4, Co=8, sh=1, sw=1, TA=1, TB=1, WT=1, OH=4, OW=4, osh=1, osw=1)
    rc = L.vince_conv_igemm(ctypes.byref(d), 0, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None, None)
    assert rc == -1 and b"Ci=6" in L.vince_last_error()
    rc = L.vince_conv_igemm(ctypes.byref(d), 7, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None, None)
    assert rc == -2
    i = InfoNCEDesc(B=32, D=96, Bk=32, K=64, frames=1, offdiag_neg=0, inv_temperature=1.0)
    assert L.vince_infonce_workspace_bytes(ctypes.byref(i)) == 0 and b"D=96" in L.vince_last_error()
    i = InfoNCEDesc(B=32, D=64, Bk=16, K=64, frames=1, offdiag_neg=0, inv_temperature=1.0)
    assert L.vince_infonce_workspace_bytes(ctypes.byref(i)) == 0   # short batch rejected (App. D item 4)
    h = ctypes.c_void_p()
    assert L.vince_trunk_create(ctypes.byref(TrunkCfg(arch=34, N=2, H=64, W=64, dtype=0)), ctypes.byref(h)) == -6


def test_trunk_plan_matches_reference_state_dict_layout(L):
    """The engine's parameter table must be the reference's state-dict order (minus fc and BN buffers)."""
    from oracle import vince_oracle as vo
    from vince_amd._lib import TrunkCfg
    for arch, name in [(18, "ResNet18"), (50, "ResNet50")]:
        h = ctypes.c_void_p()
        assert L.vince_trunk_create(ctypes.byref(TrunkCfg(arch=arch, N=2, H=64, W=64, dtype=1)), ctypes.byref(h)) == 0
        n = L.vince_trunk_num_params(h)
        got = []
        for idx in range(n):
            buf = ctypes.create_string_buffer(128)
            kind, shape, bn = ctypes.c_int32(), (ctypes.c_int32 * 4)(), ctypes.c_int32()
            assert L.vince_trunk_param_info(h, idx, buf, 128, ctypes.byref(kind), ctypes.byref(shape), ctypes.byref(bn)) == 0
            shp = tuple(shape) if kind.value == 0 else (shape[0],)
            got.append((buf.value.decode(), shp))
        want = [(nm[len("feature_extractor.model."):], tuple(shape)) for nm, shape, k in vo.trunk_spec(name)
                if k in ("conv", "bn_weight", "bn_bias")]
        assert got == want
        assert L.vince_trunk_out_channels(h) == vo.ARCH[name]["out_channels"]
        assert L.vince_trunk_workspace_bytes(h) > 0 and L.vince_trunk_weight_cache_bytes(h) > 0
        L.vince_trunk_destroy(h)


def test_product_path_refuses_cpu():
    from vince_amd import ops
    with pytest.raises(RuntimeError):
        ops.l2norm_fwd(torch.randn(4, 64))


def test_driver_build_entry_point_passes():
    """The driver's "does it build" check is __graft_entry__.build(): it must succeed against the library as built (it compared
    the ABI version with a literal once, and failed from the first bump on)."""
    import __graft_entry__ as entry
    entry.build()
