#!/usr/bin/env python
"""ISA audit for the packed-fp32 operand forms that misbehave beside f16 MFMAs on MI355X (DESIGN.md section 7,
tools/pk_f32_beside_mfma_probe.hip): a `v_pk_*_f32` whose LOW lane reads the HIGH half of src1 while src0 reads its low half
(op_sel[src1] = 1, op_sel[src0] = 0).

    python tools/isa_audit.py file.s [...]            # hipcc -S output or llvm-objdump -d output
    python tools/isa_audit.py --rccl                  # extract the gfx950 code object of torch's librccl.so and audit it
    python tools/isa_audit.py --csrc                  # compile every csrc/*.hip to gfx950 assembly and audit it

Prints one line per input: packed-fp32 instructions, op_sel histogram, fragile count (+ the kernels that hold them).
tests/test_isa_audit.py imports scan() for the csrc files; the RCCL result of the round is kept under profiles/."""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"\b(v_pk_\w+_f32)\b(.*)")
SEL = re.compile(r"op_sel:\[([01,]+)\]")
LABEL = re.compile(r"^(?:[0-9a-f]+ <)?([A-Za-z_][\w.$]*)>?:\s*$")


def scan(path):
    """-> {'pk': n, 'forms': Counter, 'fragile': [(kernel, line)], 'mfma_f16': n}"""
    forms, fragile, pk, mf = collections.Counter(), [], 0, 0
    kernel = "?"
    with open(path, errors="replace") as f:
        for line in f:
            m = LABEL.match(line)
            if m:
                kernel = m.group(1)
                continue
            if "v_mfma_" in line and ("_f16" in line or "_bf16" in line or "f8" in line):
                mf += 1
            m = PK.search(line)
            if not m:
                continue
            pk += 1
            sel = SEL.search(m.group(2))
            bits = [int(b) for b in sel.group(1).split(",")] if sel else [0, 0]
            forms["%s op_sel:%s" % (m.group(1), bits)] += 1
            if len(bits) > 1 and bits[1] == 1 and bits[0] == 0:
                fragile.append((kernel, line.strip().split("//")[0].strip()))
    return {"pk": pk, "forms": forms, "fragile": fragile, "mfma_f16": mf}


def report(name, r, out=sys.stdout):
    out.write("%s: %d packed-fp32 instructions, %d low-precision MFMA instructions, FRAGILE (op_sel[src1]=1, op_sel[src0]=0): %d\n"
              % (name, r["pk"], r["mfma_f16"], len(r["fragile"])))
    for k, v in sorted(r["forms"].items()):
        out.write("    %6d  %s\n" % (v, k))
    by = collections.Counter(k for k, _ in r["fragile"])
    for k, v in by.most_common():
        out.write("    FRAGILE x%d in %s\n" % (v, k))
    for k, l in r["fragile"][:8]:
        out.write("        %s\n" % l)


def rccl_asm(workdir):
    import torch
    lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    fat, co, asm = (os.path.join(workdir, n) for n in ("rccl_fatbin.bin", "rccl_gfx950.co", "rccl_gfx950.s"))
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    with open(asm, "w") as f:
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, stdout=f)
    return lib, asm


def csrc_asm(src, workdir):
    from sound_event_detection_dcase2017_task4_amd import build
    out = os.path.join(workdir, os.path.basename(src) + ".s")
    flags = [f for f in build.FLAGS if f != "-fPIC"]
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", src, "-o", out],
                   check=True, capture_output=True)
    return out


def main(argv):
    sys.path.insert(0, REPO)
    bad = 0
    with tempfile.TemporaryDirectory() as wd:
        if "--rccl" in argv:
            lib, asm = rccl_asm(wd)
            r = scan(asm)
            report("%s [gfx950 code object, %d bytes of disassembly]" % (lib, os.path.getsize(asm)), r)
            bad += len(r["fragile"])
        if "--csrc" in argv:
            from sound_event_detection_dcase2017_task4_amd import build
            for s in build.SOURCES:
                r = scan(csrc_asm(os.path.join(build.CSRC, s), wd))
                report("csrc/" + s, r)
                bad += len(r["fragile"])
        for p in argv:
            if not p.startswith("--"):
                r = scan(p)
                report(p, r)
                bad += len(r["fragile"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
