"""Bisecting the packed-fp32 finding at the INSTRUCTION level.  CPU only (hipcc + the LLVM assembler):

    python scripts/repro_pk_f32/make_variants.py            # -> scripts/repro_pk_f32/variants/*.hsaco + variants.json

raster.hip is compiled to gfx950 assembly twice -- with the product's -fno-slp-vectorize (no packed-fp32 instruction) and
without it (49 v_pk_{add,mul,fma}_f32 in k_raster, the build that returned wrong lanes 48-63 next to an MFMA kernel of
another stream) -- and the second listing is edited per variant: chosen packed instructions are REPLACED by the two scalar
instructions they stand for (same operands, op_sel / op_sel_hi / neg_lo / neg_hi honoured), or an `s_nop` is inserted in
front of one.  Every variant is assembled into a code object that scripts/repro_pk_f32/run_variants.py loads with
hipModuleLoad and launches in place of the library's k_raster, alone and beside a GEMM.  Nothing else differs between the
variants: same registers, same schedule, same occupancy."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "foundationpose_amd", "csrc")
OUT = os.path.join(HERE, "variants")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z8k_raster7fp_meshPKfS1_5fp_k9iiiiiffffi9RenderOut8RenderWs"
PK = re.compile(r"^\tv_pk_(add|mul|fma)_f32 (.*)$")


def device_asm(path, slp):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{os.path.join(ROOT, 'include')}", "-ffp-contract=off"]
    if not slp:
        flags.append("-fno-slp-vectorize")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "raster.hip"), "-o", path],
                          stderr=subprocess.DEVNULL)


def split_operands(rest):
    """'v[30:31], v[30:31], 0.5 op_sel_hi:[1,0]' -> (['v[30:31]', 'v[30:31]', '0.5'], {'op_sel_hi': [1, 0]})"""
    mods = {}
    for name in ("op_sel_hi", "op_sel", "neg_lo", "neg_hi"):
        m = re.search(r"\b" + name + r":\[([0-9,]+)\]", rest)
        if m:
            mods[name] = [int(v) for v in m.group(1).split(",")]
            rest = rest.replace(m.group(0), "")
    ops, depth, cur = [], 0, ""
    for ch in rest.strip():
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops, mods


def half(op, hi):
    """the 32-bit register (or constant) that is the lo / hi half of a packed operand"""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + (1 if hi else 0)}"
    return op            # inline constant / literal: the assembler repeats it (op_sel_hi = 0 is how the compiler asks for that)


def scalarise(kind, rest):
    """-> the two scalar instructions (lo, hi) equivalent to `v_pk_<kind>_f32 <rest>`, or None when register overlap makes the
    order matter"""
    ops, mods = split_operands(rest)
    dst, srcs = ops[0], ops[1:]
    n = len(srcs)
    op_sel = mods.get("op_sel", [0] * n)
    op_sel_hi = mods.get("op_sel_hi", [1] * n)
    neg_lo = mods.get("neg_lo", [0] * n)
    neg_hi = mods.get("neg_hi", [0] * n)
    lo_ops = [("-" if neg_lo[k] else "") + half(srcs[k], op_sel[k]) for k in range(n)]
    hi_ops = [("-" if neg_hi[k] else "") + half(srcs[k], op_sel_hi[k]) for k in range(n)]
    d_lo, d_hi = half(dst, 0), half(dst, 1)
    name = {"add": "v_add_f32_e64", "mul": "v_mul_f32_e64", "fma": "v_fma_f32"}[kind]
    lo = f"\t{name} {d_lo}, " + ", ".join(lo_ops)
    hi = f"\t{name} {d_hi}, " + ", ".join(hi_ops)
    lo_reads = {o.lstrip("-") for o in lo_ops}
    hi_reads = {o.lstrip("-") for o in hi_ops}
    if d_lo not in hi_reads:
        return [lo, hi]
    if d_hi not in lo_reads:
        return [hi, lo]
    return None


def kernel_span(lines):
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    return a, b


def build(name, lines):
    s = os.path.join(OUT, name + ".s")
    with open(s, "w") as f:
        f.write("\n".join(lines) + "\n")
    o = os.path.join(OUT, name + ".o")
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([f"{LLVM}/ld.lld", "-shared", o, "-o", os.path.join(OUT, name + ".hsaco")])
    os.remove(o)
    dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", os.path.join(OUT, name + ".hsaco")], text=True)
    a = dis.index(KERNEL)
    body = dis[a:dis.index("s_endpgm", a)]
    return len(re.findall(r"v_pk_(add|mul|fma)_f32", body))


def main():
    os.makedirs(OUT, exist_ok=True)
    base_s, nopk_s = os.path.join(OUT, "_base.s"), os.path.join(OUT, "_nopk.s")
    device_asm(base_s, slp=True)
    device_asm(nopk_s, slp=False)
    base = open(base_s).read().split("\n")
    a, b = kernel_span(base)
    sites = [i for i in range(a, b) if PK.match(base[i])]
    print(f"{len(sites)} packed-fp32 instructions in k_raster")
    manifest = {"sites": [dict(index=k, line=base[i].strip(), splittable=scalarise(*PK.match(base[i]).groups()) is not None)
                          for k, i in enumerate(sites)], "variants": {}}

    def variant(name, split=(), nop_before=None, note=""):
        edits = {}                                   # original line index -> replacement lines
        for k in split:
            rep = scalarise(*PK.match(base[sites[k]]).groups())
            if rep is not None:
                edits[sites[k]] = rep
        if nop_before is not None:
            k, cnt = nop_before
            edits[sites[k]] = [f"\ts_nop {cnt}"] + edits.get(sites[k], [base[sites[k]]])
        lines = list(base)
        for i in sorted(edits, reverse=True):
            lines[i:i + 1] = edits[i]
        left = build(name, lines)
        manifest["variants"][name] = dict(packed_left=left, split=sorted(split), nop_before=nop_before, note=note)
        print(f"{name}: {left} packed instructions left")

    n = len(sites)
    ok = [k for k in range(n) if manifest["sites"][k]["splittable"]]
    variant("base", note="the build that failed: all packed instructions in place")
    manifest["variants"]["nopk"] = dict(packed_left=build("nopk", open(nopk_s).read().split("\n")), note="the product's build (-fno-slp-vectorize)")
    variant("all_split", split=ok, note="every packed instruction replaced by its two scalar halves")
    variant("only_site0_split", split=[0], note="only the pixel-centre add (i + 0.5, j + 0.5) scalarised")
    variant("all_but_site0_split", split=[k for k in ok if k != 0], note="the pixel-centre add is the ONLY packed instruction left")
    variant("nop_before_site0", nop_before=(0, 3), note="s_nop 3 between the two v_cvt_f32_i32 and the packed add that reads them")
    half_n = n // 2
    variant("first_half_split", split=[k for k in ok if k < half_n], note="bisection")
    variant("second_half_split", split=[k for k in ok if k >= half_n], note="bisection")
    # the first cluster (barycentric set-up: sites 1..10) and the rest, each on its own
    variant("sites_1_10_split", split=[k for k in ok if 1 <= k <= 10])
    variant("sites_11_end_split", split=[k for k in ok if k >= 11])
    # ---- second round: all_split still fails with ONE packed-fp32 instruction left (site 9, whose halves swap registers) -- and one
    # v_pk_mov_b32.  Hand-written replacements for both, alone and together (v62 / v63 are inside the kernel's allocation granule:
    # 62 registers requested, 64 allocated, never touched by the compiler's code).
    def variant2(name, split, site9, pkmov, extra_edit=None, note=""):
        edits = {}
        for k in split:
            rep = scalarise(*PK.match(base[sites[k]]).groups())
            if rep is not None:
                edits[sites[k]] = rep
        if site9:
            assert base[sites[9]].strip() == "v_pk_mul_f32 v[30:31], v[32:33], v[30:31] op_sel:[0,1] op_sel_hi:[1,0]", base[sites[9]]
            edits[sites[9]] = ["\tv_mul_f32_e64 v63, v32, v31", "\tv_mul_f32_e64 v31, v33, v30", "\tv_mov_b32_e32 v30, v63"]
        if pkmov:
            i = next(j for j in range(a, b) if base[j].strip() == "v_pk_mov_b32 v[34:35], v[36:37], v[34:35] op_sel:[1,0]")
            edits[i] = ["\tv_mov_b32_e32 v35, v34", "\tv_mov_b32_e32 v34, v37"]      # dst.lo = src0.hi, dst.hi = src1.lo
        if extra_edit:
            edits.update(extra_edit(edits))
        lines = list(base)
        for i in sorted(edits, reverse=True):
            lines[i:i + 1] = edits[i]
        left = build(name, lines)
        dis_left = sum(1 for l in lines[a:] if l.strip().startswith("v_pk_"))
        manifest["variants"][name] = dict(packed_left=left, any_v_pk_left=dis_left, note=note)
        print(f"{name}: {left} packed-fp32 left, {dis_left} v_pk_* of any kind in the listing")
    variant2("zero_pk", ok, True, True, note="no packed instruction of any kind: the SLP build's registers and schedule, scalar arithmetic")
    variant2("only_site9_packed", ok, False, True, note="v_pk_mul_f32 v[30:31], v[32:33], v[30:31] op_sel:[0,1] op_sel_hi:[1,0] is the only packed instruction")
    variant2("only_pk_mov_packed", ok, True, False, note="v_pk_mov_b32 v[34:35], v[36:37], v[34:35] op_sel:[1,0] is the only packed instruction")
    variant2("base_site9_split", [], True, False, note="all other 41 packed-fp32 instructions (and v_pk_mov_b32) in place, only site 9 scalarised")

    def site9_other_dst(edits):
        return {sites[9]: ["\tv_pk_mul_f32 v[62:63], v[32:33], v[30:31] op_sel:[0,1] op_sel_hi:[1,0]", "\tv_mov_b32_e32 v30, v62", "\tv_mov_b32_e32 v31, v63"]}
    variant2("only_site9_packed_other_dst", ok, False, True, extra_edit=site9_other_dst,
             note="site 9 is the only packed instruction, but writes v[62:63] (copied back by two v_mov_b32): the crossing without the in-place overlap")
    # third round: where does the hazard sit relative to the one packed instruction?  Idle states in front of it / behind it
    NOPS = ["\ts_nop 7"] * 4
    def nops_before(edits):
        return {sites[9]: NOPS + [base[sites[9]]]}
    def nops_after(edits):
        return {sites[9]: [base[sites[9]]] + NOPS}
    def nops_both(edits):
        return {sites[9]: NOPS + [base[sites[9]]] + NOPS}
    def drain_before(edits):
        return {sites[9]: ["\ts_waitcnt vmcnt(0) lgkmcnt(0)"] + NOPS + [base[sites[9]]]}
    variant2("only_site9_packed_nops_before", ok, False, True, extra_edit=nops_before, note="4 x s_nop 7 in front of the one packed instruction")
    variant2("only_site9_packed_nops_after", ok, False, True, extra_edit=nops_after, note="4 x s_nop 7 behind the one packed instruction")
    variant2("only_site9_packed_nops_both", ok, False, True, extra_edit=nops_both, note="4 x s_nop 7 on either side")
    variant2("only_site9_packed_drained", ok, False, True, extra_edit=drain_before, note="s_waitcnt vmcnt(0) lgkmcnt(0) + 4 x s_nop 7 in front")
    variant2("base_pk_mov_split", [], False, True, note="all 42 packed-fp32 instructions in place, only v_pk_mov_b32 replaced by two v_mov_b32")
    with open(os.path.join(OUT, "variants.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    for fn in os.listdir(OUT):
        if fn.endswith(".s") and not fn.startswith("_"):
            os.remove(os.path.join(OUT, fn))


if __name__ == "__main__":
    sys.exit(main())
