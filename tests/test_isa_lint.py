"""tools/isa_lint.py: the hazards it exists for are found in hand-made instruction lists (CPU test: no GPU, no compiler).

The lint runs over the device ISA of the built library in __graft_entry__.build(); this test pins WHAT it flags, so that a rule
that silently stops matching (a changed disassembly format, say) shows up here and not as a wrong consensus on the device."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def _insts(text):
    return [p for p in (isa_lint.parse(line) for line in text.strip().split("\n")) if p]


def test_store_data_hazard_is_found_and_an_s_nop_clears_it():
    bad = _insts("""
        global_store_dwordx3 v51, v[6:8], s[0:1]                   // 000000001234: DC7C0000
        v_perm_b32 v6, v55, v56, s68
    """)
    assert any(f.startswith("R1") for f in isa_lint.lint_kernel("k", bad))
    ok = _insts("""
        global_store_dwordx3 v51, v[6:8], s[0:1]
        s_nop 1
        v_perm_b32 v6, v55, v56, s68
    """)
    assert isa_lint.lint_kernel("k", ok) == []
    # a buffer store's data is its FIRST operand; two unrelated instructions are two wait states
    assert any(f.startswith("R1") for f in isa_lint.lint_kernel("k", _insts("""
        buffer_store_dwordx4 v[6:9], v49, s[72:75], s20 offen
        s_add_i32 s18, s18, 1
        v_pk_max_u16 v8, v8, v7
    """)))
    assert isa_lint.lint_kernel("k", _insts("""
        buffer_store_dwordx4 v[6:9], v49, s[72:75], s20 offen
        s_add_i32 s18, s18, 1
        s_cmp_eq_u32 s18, s19
        v_pk_max_u16 v8, v8, v7
    """)) == []
    # 64-bit stores have no such hazard
    assert isa_lint.lint_kernel("k", _insts("""
        global_store_dwordx2 v51, v[6:7], s[0:1]
        v_mov_b32_e32 v6, 0
    """)) == []


def test_dpp_read_behind_a_vector_write_is_found():
    bad = _insts("""
        v_pk_max_u16 v54, v50, v53 op_sel:[0,1] op_sel_hi:[1,1]
        v_max_u32_dpp v50, v54, v54 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
    """)
    assert any(f.startswith("R2") for f in isa_lint.lint_kernel("k", bad))
    ok = _insts("""
        v_pk_max_u16 v54, v50, v53 op_sel:[0,1] op_sel_hi:[1,1]
        s_nop 1
        v_max_u32_dpp v50, v54, v54 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
    """)
    assert isa_lint.lint_kernel("k", ok) == []


def test_lane_select_fresh_from_a_vector_instruction_is_found():
    bad = _insts("""
        v_readlane_b32 s16, v2, s3
        v_readlane_b32 s18, v2, s16
    """)
    assert any(f.startswith("R3") for f in isa_lint.lint_kernel("k", bad))
    assert isa_lint.lint_kernel("k", _insts("""
        v_readlane_b32 s16, v2, s3
        s_nop 3
        v_readlane_b32 s18, v2, s16
    """)) == []
