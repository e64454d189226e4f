"""CPU: libmmda_hip.so loads and exports every symbol include/mmda_hip.h declares; the ctypes table covers them all.
No compute calls (no GPU here)."""
import ctypes
import os
import re

from mmda_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mmda_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmda_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in mmda_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in mmda_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"


def test_struct_sizes_match_c_layout():
    """ctypes mirrors of the argument structs must have the C compiler's size (catches field drift)."""
    import subprocess, tempfile, textwrap
    code = textwrap.dedent("""
        #include <stdio.h>
        #include "mmda_hip.h"
        int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mmda_gemm_args), sizeof(mmda_ln_args), sizeof(mmda_ln_bwd_args),
                          sizeof(mmda_lstm_desc), sizeof(mmda_misa_config), sizeof(mmda_gemm_bf16_args),
                          sizeof(mmda_convert_job), sizeof(mmda_skinny_args), sizeof(mmda_transpose_job), sizeof(mmda_gru_pad_job),
                          sizeof(mmda_mx8_quant_job), sizeof(mmda_mx8_args), sizeof(mmda_act_params));
                   return 0;}""")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(code)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], check=True, capture_output=True, text=True).stdout.split()
    sizes = [ctypes.sizeof(x) for x in (_lib.GemmArgs, _lib.LnArgs, _lib.LnBwdArgs, _lib.LstmDesc, _lib.MisaConfig,
                                       _lib.GemmBf16Args, _lib.ConvertJob, _lib.SkinnyArgs, _lib.TransposeJob, _lib.GruPadJob,
                                       _lib.Mx8QuantJob, _lib.Mx8Args, _lib.ActParams)]
    assert [int(x) for x in out] == sizes


def test_error_codes_without_gpu():
    lib = _lib.load()
    assert lib.mmda_abi_version() == 1
    assert lib.mmda_gemm(None, None) == -1                      # MMDA_EINVAL, no launch attempted
    assert lib.mmda_lstm_packed_bytes(_lib.BF16, 300, 0) == 19 * 4 * 10 * 64 * 16
    assert lib.mmda_lstm_packed_bytes(_lib.BF16, 300, 1) == 19 * 38 * 64 * 16
    assert lib.mmda_lstm_packed_bytes(_lib.F32, 35, 0) == 3 * 4 * 3 * 64 * 16
    assert lib.mmda_lstm_packed_bytes(_lib.F32, 0, 0) == -1
