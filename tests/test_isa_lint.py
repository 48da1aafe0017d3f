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


def test_lint_checks_the_last_workgroup_handshake():
    """tools/isa_lint.py also reads the classifier + counters kernel's cross-workgroup hand-shake (round 4 found the
    workgroup-scope release fence compiled to nothing): sc1 key stores / loads and an `s_waitcnt vmcnt(0)` of its own in front
    of the release barrier."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    good = """
0000000000001000 <_ZN12_GLOBAL__N_118pwconv_rows_kernelILb1EEEvPKf>:
	global_store_dwordx2 v[4:5], v[2:3], off sc1
	s_cbranch_vccz 252
	s_waitcnt vmcnt(0)
	s_cbranch_execnz 5
	buffer_wbl2 sc1
	s_waitcnt vmcnt(0)
	buffer_inv sc1
	s_barrier
	global_load_dwordx2 v[4:5], v[0:1], off sc1
"""
    assert isa_lint.lint_handshake(good) == []
    no_wait = good.replace("	s_cbranch_vccz 252\n	s_waitcnt vmcnt(0)\n", "	s_cbranch_vccz 252\n")
    assert any("vmcnt(0)" in p for p in isa_lint.lint_handshake(no_wait))
    no_sc1 = good.replace("off sc1\n	s_cbranch", "off\n	s_cbranch")
    assert any("key store" in p for p in isa_lint.lint_handshake(no_sc1))
    assert isa_lint.lint_handshake("0000000000001000 <other_kernel>:\n	s_endpgm\n") is None
