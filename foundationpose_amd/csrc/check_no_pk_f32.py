"""Build-time check (csrc/Makefile, target `check`): no gfx950 code object of the library may contain a packed-fp32 VALU
instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  On MI355X / ROCm 7.2 the rasteriser built WITH them returned wrong
lanes while an MFMA kernel of another stream shared the chip; the library is compiled with
-fno-slp-vectorize, and this script fails the build if a toolchain change or a new kernel brings them back.
Usage: python check_no_pk_f32.py libfp_amd.so"""
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fatbin", lib, f"{d}/copy.so"])
        blob = open(f"{d}/fatbin", "rb").read()
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz > 0:
                yield blob[i + o:i + o + sz]
        pos = i + 24


def main(lib):
    bad, n = [], 0
    for co in code_objects(lib):
        n += 1
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            asm = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f.name]).decode()
        kernel = "?"
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
            elif re.search(r"\bv_pk_(add|mul|fma)_f32\b", line):
                bad.append((kernel, line.strip()))
    if n == 0:
        sys.exit(f"{lib}: no gfx950 code object found")
    if bad:
        for k, l in bad[:20]:
            print(f"  {k}: {l}", file=sys.stderr)
        sys.exit(f"{lib}: {len(bad)} packed-fp32 VALU instructions in gfx950 code (see DESIGN.md 3.5)")
    print(f"{lib}: {n} gfx950 code objects, no packed-fp32 VALU instruction")


if __name__ == "__main__":
    main(sys.argv[1])
