"""Build-time guard for the gfx950 store-data hazard (profiles/r3_dw_flat_race.txt): no 16-byte buffer store of the built
library may be followed within two wait states by a vector instruction that writes one of its data registers - hipcc does
not guard that form when soffset is a register, and on MI355X the write then overtakes the store's data read under load."""
import os
import subprocess
import sys

from conftest import ROOT


def test_no_unguarded_write_of_wide_store_data():
    lib = os.path.join(ROOT, "quantization", "mxnet_amd", "csrc", "libfakequant.so")
    assert os.path.exists(lib), "library not built"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), lib], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 unprotected" in r.stdout


def test_lint_recognises_the_pattern():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    bad = """
0000000000001000 <kern>:
	buffer_store_dwordx4 v[4:7], v8, s[36:39], s52 offen
	v_cndmask_b32_e32 v4, v97, v74, vcc
	s_waitcnt lgkmcnt(0)
"""
    ok = bad.replace("v_cndmask_b32_e32 v4", "s_nop 1\n\tv_cndmask_b32_e32 v4")
    far = bad.replace("v_cndmask_b32_e32 v4", "v_cndmask_b32_e32 v9")
    assert len(isa_lint.lint_text(bad)) == 1
    assert isa_lint.lint_text(ok) == [] and isa_lint.lint_text(far) == []
