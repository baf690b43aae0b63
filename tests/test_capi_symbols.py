"""CPU: libnele_hip.so loads (no GPU needed) and exports every symbol include/nele_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'nele_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nele_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for must in ('nele_version', 'nele_last_error_string', 'nele_stft_band', 'nele_imcra_band', 'nele_gain_istft'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, 'nele_gan_amd', 'libnele_hip.so')
    assert os.path.exists(path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in include/nele_hip.h but not exported: %s" % missing
    lib.nele_version.restype = ctypes.c_int
    assert lib.nele_version() >= 100


def test_python_binding_covers_header():
    import nele_gan_amd
    from nele_gan_amd import _lib
    nele_gan_amd.load_all_bindings()
    bound = set(_lib._SIGS) | {'nele_version', 'nele_last_error_string'}
    assert set(declared_symbols()) <= bound, sorted(set(declared_symbols()) - bound)


def test_invalid_arguments_are_reported_without_a_gpu():
    from nele_gan_amd import _lib
    st = _lib.lib.nele_stft_band(None, 0, 0, 0.0, None, None, None)
    assert st == -1
    assert b'nele_stft_band' in _lib.lib.nele_last_error_string()


def test_product_library_has_no_switches_and_the_test_library_exports_the_same_abi():
    """libnele_hip.so (product): NELE_* environment switches are compile-time constants, the superseded kernel variants are not built.
    libnele_hip_ab.so (tests / tools, -DNELE_AB): same sources, same exported symbols, switches read from the environment."""
    prod = ctypes.CDLL(os.path.join(ROOT, 'nele_gan_amd', 'libnele_hip.so'))
    ab_path = os.path.join(ROOT, 'nele_gan_amd', 'libnele_hip_ab.so')
    assert os.path.exists(ab_path), "make -C nele_gan_amd/csrc builds both libraries"
    ab = ctypes.CDLL(ab_path)
    assert prod.nele_build_has_ab_switches() == 0 and ab.nele_build_has_ab_switches() == 1
    missing = [s for s in declared_symbols() if not hasattr(ab, s)]
    assert not missing, missing
    # no getenv in the product library's import table
    import subprocess
    nm = subprocess.run(['nm', '-D', '--undefined-only', os.path.join(ROOT, 'nele_gan_amd', 'libnele_hip.so')], stdout=subprocess.PIPE, text=True).stdout
    assert 'getenv' not in nm
    nm_ab = subprocess.run(['nm', '-D', '--undefined-only', ab_path], stdout=subprocess.PIPE, text=True).stdout
    assert 'getenv' in nm_ab


def test_plan_operation_table_is_generated_from_the_header():
    """csrc/plan_ops.inc (the dispatch table of nele_plan_run) is generated from include/nele_hip.h by tools/gen_plan_ops.py."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_plan_ops.py'), '--check'])
    assert r.returncode == 0, "include/nele_hip.h changed: run python tools/gen_plan_ops.py and rebuild"
    from nele_gan_amd import _lib
    for name in ('nele_conv16', 'nele_glayer16_fwd', 'nele_cln_bwd', 'nele_event_record', 'nele_vec_add', 'nele_gap_mlp_bwd_var16'):
        op = _lib.lib.nele_plan_op_id(name.encode())
        assert op >= 0 and _lib.lib.nele_plan_op_nargs(op) == len(_lib._SIGS[name]), name
    assert _lib.lib.nele_plan_op_id(b'nele_version') == -1
