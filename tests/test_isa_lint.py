"""The shipped HIP library must be free of the hipcc code-generation defect that made k_step_hybrid2<7,3> fault in
round 2: VGPR rematerialisations placed in front of the EXEC restore of an `s_cbranch_execz` join block,
behind SGPR-spill writelanes.  tools/isa_lint.py disassembles every gfx950 kernel of librgs_hip.so (no GPU needed) and
checks the rule; the lint itself is pinned on the block that faulted (profiles/r03/fault_rootcause/)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

LIB = os.path.join(ROOT, "robo-game-sim_amd", "csrc", "librgs_hip.so")


def test_lint_flags_the_block_that_faulted(tmp_path):
    """The join block of commit 941728a's k_step_hybrid2<7,3>, as the compiler emitted it: four constants are
    materialised before `s_or_b64 exec`; taken with EXEC = 0 they are never written."""
    src = open(os.path.join(ROOT, "profiles", "r03", "fault_rootcause", "isa_join_block_LBB12_253.s")).read()
    f = tmp_path / "block.s"
    f.write_text("_ZN4gsim14k_step_hybrid2ILi7ELi3EEEvNS_8StepArgsE:\n" + src)
    found = isa_lint.lint(str(f))
    assert len(found) == 1 and found[0][1] == ".LBB12_253"
    assert [b.split(": ", 1)[1] for b in found[0][2]] == ["v_mov_b32_e32 v17, 0x60", "v_mov_b32_e32 v16, 0xa0",
                                                          "v_mov_b32_e32 v250, 0x50", "v_mov_b32_e32 v252, 0x88"]


def test_lint_accepts_a_correct_join(tmp_path):
    f = tmp_path / "ok.s"
    f.write_text("_Zk:\n\ts_and_saveexec_b64 s[0:1], vcc\n\ts_cbranch_execz .LBB0_2\n\tv_mov_b32_e32 v1, 2\n.LBB0_2:\n"
                 "\tv_writelane_b32 v9, s4, 0\n\ts_or_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v2, 0x88\n\ts_endpgm\n")
    assert isa_lint.lint(str(f)) == []


@pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_shipped_library_is_clean():
    assert os.path.exists(LIB), "build the library first (make -C robo-game-sim_amd/csrc)"
    found = isa_lint.lint(LIB)
    assert found == [], "hipcc defect pattern in: %s" % [(k, w) for k, w, _ in found]
