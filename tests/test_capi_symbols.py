"""CPU: libemsanet_hip.so loads (no GPU needed: no compute call is made), exports every symbol
include/emsanet_hip.h declares, and the ctypes signatures have the declared arity."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'emsanet_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|int64_t|const char\*)\s+(emsa_\w+)\s*\(([^;]*?)\)\s*;', src, re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_header_symbols_exported_and_bound():
    import __graft_entry__ as g
    from emsanet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    decls = _declared()
    assert len(decls) >= 40, decls
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: arity differs from the header"
    for name in _lib.SIGNATURES:
        assert name in decls, f"{name} bound in _lib.py but not declared in the header"


def test_library_identity_without_gpu():
    from emsanet_amd import _lib
    L = _lib.lib()
    assert L.emsa_arch() == b'gfx950'
    assert L.emsa_version() >= 1
    # rows to allocate = per-workgroup partial rows + 16 slice-sum rows
    assert L.emsa_bn_bwd_rows(10 ** 6, 64) == 1024 + 16 and L.emsa_bn_bwd_rows(1, 64) == 1 + 16
    assert L.emsa_prof_name(0).startswith(b'conv_igemm_kernel')


def test_geometry_rejected_cleanly():
    """argument validation runs on the host before any launch"""
    from emsanet_amd import _lib
    from emsanet_amd._lib import EmsaConvGeom
    L = _lib.lib()
    g = EmsaConvGeom(1, 8, 8, 8, 8, 6, 8, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 384, 48, 6, 8)   # k_ch % 4
    assert L.emsa_conv_stats_rows(ctypes.byref(g)) == -1
    assert L.emsa_conv_igemm(ctypes.byref(g), None, None, None, None, None, None, None, None, 0,
                             None, 0, 0, None) == -1
    assert L.emsa_bn_act_fwd(None, None, None, None, None, None, 1, 1, 64, 0, None, None) == -2


def test_gfx950_code_object_present():
    """the shared object embeds a gfx950 code object (hipcc --offload-arch=gfx950)"""
    from emsanet_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob
