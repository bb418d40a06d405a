"""tools/isa_count.py: the assembly parser behind the static instruction model (profiles/r01_static_model_d.txt)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import isa_count  # noqa: E402

ASM = """
	.text
_ZN3pbc4leafEv:                         ; @_ZN3pbc4leafEv
; %bb.0:
	s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)
	v_mad_u64_u32 v[0:1], s[0:1], v2, v3, 0
	v_mad_i64_i32 v[4:5], s[0:1], v2, v3, v[0:1]
	v_add_u32_e32 v0, v0, v1
	ds_read_b32 v6, v31
	scratch_load_dword v7, off, s32
	s_setpc_b64 s[30:31]
.Lfunc_end0:
	.size	_ZN3pbc4leafEv, .Lfunc_end0-_ZN3pbc4leafEv
_Z6kernelPh:                            ; @_Z6kernelPh
; %bb.0:
	s_getpc_b64 s[4:5]
	s_add_u32 s4, s4, _ZN3pbc4leafEv@rel32@lo+4
	s_addc_u32 s5, s5, _ZN3pbc4leafEv@rel32@hi+12
.LBB1_1:                                ; =>This Inner Loop Header: Depth=1
	v_mov_b32_e32 v0, 0
	s_swappc_b64 s[30:31], s[4:5]
	global_store_dword v1, v0, s[0:1]
	s_cbranch_scc1 .LBB1_1
; %bb.2:
	s_endpgm
.Lfunc_end1:
"""


def test_isa_count_classifies_and_splits_blocks(tmp_path):
    p = tmp_path / "t.s"
    p.write_text(ASM)
    funcs = isa_count.parse(str(p))
    assert list(funcs) == ["_ZN3pbc4leafEv", "_Z6kernelPh"]
    leaf = isa_count.totals(funcs["_ZN3pbc4leafEv"])
    assert leaf == {"salu": 2, "mad64": 2, "valu": 1, "lds": 1, "vmem": 1}
    kern = funcs["_Z6kernelPh"]
    assert list(kern) == ["entry", ".LBB1_1"]
    assert isa_count.totals({"b": kern[".LBB1_1"]}) == {"valu": 1, "salu": 3, "vmem": 1}   # blocks end at labels: s_endpgm belongs to the last one
