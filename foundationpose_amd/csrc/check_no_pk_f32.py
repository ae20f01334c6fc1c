"""Build-time check (csrc/Makefile, target `check`): no gfx950 code object of the library may contain a packed-fp32 VALU
instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  On MI355X / ROCm 7.2 the rasteriser built WITH them returned wrong
lanes while an MFMA kernel of another stream shared the chip (a v_pk_*_f32 with op_sel = 1 on a VGPR source reads
zero in lanes 48-63 while another wave issues independent MFMAs back to back); the library is compiled with -fno-slp-vectorize,
and this script fails the build if a toolchain change or a new kernel brings them back.
Usage: python check_no_pk_f32.py libfp_amd.so
The LLVM tools are looked for next to $HIPCC, under $ROCM_PATH, /opt/rocm and on PATH.  Round 5 (the advisor's finding): when the
tools are missing, the fat binary cannot be read, or no gfx950 code object is found in it, the check FAILS -- a silent skip let a
toolchain change switch the guard off unnoticed, and the hazard is silent wrong lanes.  The Makefile links with
--no-offload-compress so that the bundle is always in the layout read here.  A build elsewhere that knowingly goes without the
check sets FP_AMD_ALLOW_UNCHECKED=1 (then: a warning and exit 0)."""
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def find_llvm():
    cands = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin"), os.path.join(root, "llvm", "bin")]
    for r in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if r:
            cands.append(os.path.join(r, "lib", "llvm", "bin"))
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "llvm-objcopy")):
            return c
    if shutil.which("llvm-objdump") and shutil.which("llvm-objcopy"):
        return os.path.dirname(shutil.which("llvm-objdump"))
    return None


LLVM = find_llvm()


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


def unchecked(msg):
    if os.environ.get("FP_AMD_ALLOW_UNCHECKED") == "1":
        print(f"WARNING: {msg}: packed-fp32 check skipped (FP_AMD_ALLOW_UNCHECKED=1)", file=sys.stderr)
        return
    sys.exit(f"FAIL: {msg}: the packed-fp32 check could not run; fix the toolchain lookup, or set FP_AMD_ALLOW_UNCHECKED=1 to build "
             f"without it (DESIGN.md 3.5: v_pk_*_f32 returns wrong lanes next to MFMA kernels on other streams)")


def main(lib):
    if LLVM is None:
        return unchecked(f"{lib}: llvm-objdump / llvm-objcopy not found (HIPCC, ROCM_PATH, /opt/rocm, PATH)")
    bad, n = [], 0
    try:
        objs = list(code_objects(lib))
    except (subprocess.CalledProcessError, struct.error, OSError) as e:
        return unchecked(f"{lib}: cannot read the fat binary ({e})")
    if not objs:
        blob_hint = "compressed offload bundle (CCOB)?" if b"CCOB" in open(lib, "rb").read() else "no .hip_fatbin bundle of the known layout"
        return unchecked(f"{lib}: no gfx950 code object found ({blob_hint}; link with --no-offload-compress)")
    for co in objs:
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
    if bad:
        for k, l in bad[:20]:
            print(f"  {k}: {l}", file=sys.stderr)
        sys.exit(f"{lib}: {len(bad)} packed-fp32 VALU instructions in gfx950 code (see DESIGN.md 3.5)")
    print(f"{lib}: {n} gfx950 code objects, no packed-fp32 VALU instruction")


if __name__ == "__main__":
    main(sys.argv[1])
