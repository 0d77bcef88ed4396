"""ISA audit of EVERY kernel source (csrc/*.hip): none may contain a packed-fp32 instruction whose low lane reads the HIGH half
of src1 while src0 reads its low half (op_sel[src1] = 1, op_sel[src0] = 0).  On MI355X such an instruction returns wrong values
now and then while a wave of another kernel on the same CU executes v_mfma_f32_32x32x16_f16 (tools/pk_f32_beside_mfma_probe.hip,
DESIGN.md section 7) -- and with side-stream weight gradients, RCCL reductions beside the backward pass or ranks sharing a GPU,
any kernel can end up beside one.  The same scan over RCCL's own gfx950 code object (`python tools/isa_audit.py --rccl`) is
committed per round under profiles/ (clean: 0 of its 325 packed-fp32 instructions use op_sel at all)."""
import importlib.util
import os
import shutil

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

from sound_event_detection_dcase2017_task4_amd import build   # noqa: E402

_spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(REPO, "tools", "isa_audit.py"))
isa_audit = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(isa_audit)


def test_every_source_is_audited():
    assert sorted(build.SOURCES) == sorted(f for f in os.listdir(build.CSRC) if f.endswith(".hip"))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", build.SOURCES)
def test_no_fragile_packed_fp32_forms(src, tmp_path):
    r = isa_audit.scan(isa_audit.csrc_asm(os.path.join(build.CSRC, src), str(tmp_path)))
    assert not r["fragile"], "%s: %d packed-fp32 instructions with op_sel[src1] = 1, op_sel[src0] = 0, e.g. %s" % (
        src, len(r["fragile"]), r["fragile"][:2])


def test_scanner_recognises_the_fragile_form(tmp_path):
    p = tmp_path / "x.s"
    p.write_text("_Zk:\n\tv_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n"
                 "\tv_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[1,0] op_sel_hi:[1,0]\n\tv_pk_fma_f32 v[0:1], v[2:3], v[2:3], v[4:5]\n")
    r = isa_audit.scan(str(p))
    assert r["pk"] == 3 and len(r["fragile"]) == 1 and r["fragile"][0][0] == "_Zk"
