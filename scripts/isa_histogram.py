#!/usr/bin/env python3
"""Static instruction histogram of the gfx950 code of the solver kernels.

    python scripts/isa_histogram.py [--asm /tmp/seqik_hip.s] [--out profiles/r03_fused_isa.json] [--pmc profiles/traffic_r03.json]

Compiles csrc/seqik_hip.hip to device assembly (`hipcc --cuda-device-only -S`, the same flags as the library build;
no GPU needed), cuts it into kernels and classifies every instruction:

  f64_fma / f64_mul / f64_add      the three classes SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 count
  f64_trans                        v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 (quarter rate)
  f64_div_fixup                    v_div_scale_f64 / v_div_fmas_f64 / v_div_fixup_f64: the IEEE-division scaffolding
  f64_other                        v_cmp_*_f64, v_max/min_f64, v_ldexp_f64, v_rndne_f64, v_cvt_*_f64, v_frexp_*, v_trig_preop ...
  cndmask / mov / int / cmp_int    selects, moves (incl. DPP / readlane), 32/64-bit integer VALU, integer compares
  salu / smem / branch             scalar ALU (EXEC bookkeeping), scalar loads, s_cbranch / s_branch
  scratch / global / lds / waitcnt scratch (spill) accesses, global memory, ds_*, s_waitcnt

The stage bodies are identical in the one-launch kernel (seqik_fused_kernel) and in the per-stage kernels
(seqik_stage_kernel<STAGE, ...>), so the per-stage columns come from the latter.  With --pmc the static mix is set
beside the measured per-launch counters (SQ_INSTS_VALU, the F64 classes) of the fused kernel.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sequential-inverse-kinematics_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--cuda-device-only", "-S"]


def classify(op):
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "global"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        if op.startswith(("v_fma_f64", "v_fmac_f64")):
            return "f64_fma"
        if op.startswith("v_mul_f64"):
            return "f64_mul"
        if op.startswith("v_add_f64"):
            return "f64_add"
        if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
            return "f64_trans"
        if op.startswith(("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")):
            return "f64_div_fixup"
        if "_f64" in op:
            return "f64_other"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap", "v_perm", "v_bfi")):
            return "mov"
        if op.startswith("v_cmp"):
            return "cmp_int"
        return "int"
    return "other"


def demangle_hint(name):
    """Readable tag for the kernels of interest without a demangler."""
    m = re.search(r"seqik_stage_kernelILi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    if m:
        s, fk, dg, fa, ho = m.groups()
        return f"stage_kernel<{s},fk={fk},diag={dg},from_angles={fa},handoff={ho}>"
    m = re.search(r"seqik_fused_kernelILb(\d)E", name)
    if m:
        return f"fused_kernel<fk={m.group(1)}>"
    m = re.search(r"seqik_pipe_kernelILb(\d)ELi(\d)E", name)
    if m:
        return f"pipe_kernel<fk={m.group(1)},wpe={m.group(2)}>"
    m = re.search(r"seqik_chunk_pipe_kernelILb(\d)ELi(\d)ELi(\d)E", name)
    if m:
        return f"chunk_pipe_kernel<fk={m.group(1)},mode={m.group(2)},wpe={m.group(3)}>"
    m = re.search(r"seqik_chunk_kernelILb(\d)ELi(\d)E", name)
    if m:
        return f"chunk_kernel<fk={m.group(1)},mode={m.group(2)}>"
    m = re.search(r"seqik_generic_kernelILb(\d)ELb(\d)E", name)
    if m:
        return f"generic_kernel<diag={m.group(1)},grouped={m.group(2)}>"
    return name


def parse(asm_path):
    kernels = {}
    cur, hist, meta = None, None, {}
    label = re.compile(r"^(_Z[\w$.]+):")
    inst = re.compile(r"^\s+([a-z][a-z0-9_]+)(?:\s|$)")
    for line in open(asm_path):
        m = label.match(line)
        if m:
            cur = m.group(1)
            hist = collections.Counter()
            kernels[cur] = {"hist": hist}
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None if line.startswith(".Lfunc_end") else cur
            continue
        m = inst.match(line)
        if m and not line.lstrip().startswith("."):
            hist[classify(m.group(1))] += 1
            hist["_ops:" + m.group(1)] += 1
    # resource usage from the metadata comments that follow each function
    txt = open(asm_path).read()
    for name in kernels:
        i = txt.find(f"\n{name}:")
        j = txt.find("; NumVgprs:", i)
        seg = txt[j - 400:j + 900] if j > 0 else ""
        for key in ("NumVgprs", "NumAgprs", "NumSgprs", "ScratchSize", "Occupancy", "codeLenInByte", "LDSByteSize"):
            m = re.search(r"; %s: (\d+)" % key, seg)
            if m:
                kernels[name][key] = int(m.group(1))
        m = re.search(r"\.vgpr_spill_count:\s+(\d+)", txt[i:i + 2_000_000]) if i > 0 else None
    return kernels


# 32-bit vector instructions that issue in ~2.5 cycles per wavefront (measured: scripts/microbench/valu_issue.hip ->
# profiles/r03_valu_issue_costs.json); everything else -- all f64 instructions incl. compares / max / ldexp / fix-ups,
# v_cndmask_b32 with its mask in a scalar register pair, compares that write a scalar mask, 64-bit moves / integer
# instructions -- takes ~4.2 cycles, v_rcp_f64 / v_rsq_f64 ~16
TWO_CYCLE_OPS = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32",
                 "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32", "v_bfe_u32", "v_and_or_b32", "v_or3_b32", "v_add3_u32",
                 "v_lshl_add_u32", "v_lshl_or_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_min_u32", "v_max_u32", "v_bfrev_b32")


def summary(hist):
    cls = {k: v for k, v in hist.items() if not k.startswith("_ops:")}
    total = sum(cls.values())
    valu = sum(v for k, v in cls.items() if k.startswith("f64_") or k in ("cndmask", "mov", "int", "cmp_int"))
    f64_arith = sum(cls.get(k, 0) for k in ("f64_fma", "f64_mul", "f64_add", "f64_trans"))
    top = sorted(((k[5:], v) for k, v in hist.items() if k.startswith("_ops:")), key=lambda kv: -kv[1])[:40]
    two = sum(v for k, v in hist.items() if k.startswith("_ops:v_") and k[5:].split("_e")[0] in TWO_CYCLE_OPS)
    other = valu - f64_arith
    return {"total": total, "valu": valu, "f64_counted_by_pmc": f64_arith, "valu_not_f64_arith": other,
            "valu_not_f64_arith_issue_split": {"about_2.5_cycles": two, "about_4.2_cycles": other - two,
                                               "share_about_4.2_cycles": (other - two) / other if other else None},
            "classes": dict(sorted(cls.items(), key=lambda kv: -kv[1])), "top_opcodes": dict(top)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None, help="existing assembly file (default: compile csrc/seqik_hip.hip)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_fused_isa.json"))
    ap.add_argument("--pmc", default=None, help="profiles/traffic_rNN.json to set beside the static mix")
    ap.add_argument("--define", action="append", default=[], help="extra -D for the compile (A/B builds)")
    a = ap.parse_args()
    asm = a.asm
    if not asm:
        asm = "/tmp/seqik_hip_isa.s"
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-D" + d for d in a.define] + ["-o", asm, os.path.join(CSRC, "seqik_hip.hip")]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    kernels = parse(asm)
    out = {"source": "hipcc " + " ".join(FLAGS) + " csrc/seqik_hip.hip (static count of the emitted gfx950 instructions)",
           "kernels": {}}
    want = ("fused_kernel<fk=1>", "stage_kernel<1,fk=0,diag=0,from_angles=0,handoff=1>",
            "stage_kernel<2,fk=1,diag=0,from_angles=0,handoff=1>", "stage_kernel<3,fk=1,diag=0,from_angles=0,handoff=1>",
            "stage_kernel<4,fk=1,diag=0,from_angles=0,handoff=0>", "pipe_kernel<fk=1,wpe=3>", "pipe_kernel<fk=1,wpe=2>",
            "chunk_kernel<fk=1,mode=0>", "generic_kernel<diag=0,grouped=1>", "generic_kernel<diag=0,grouped=0>")
    for name, k in kernels.items():
        tag = demangle_hint(name)
        if tag in want:
            s = summary(k["hist"])
            s.update({key: k[key] for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "codeLenInByte") if key in k})
            out["kernels"][tag] = s
    if a.pmc and os.path.exists(a.pmc):
        t = json.load(open(a.pmc))
        n = t.get("fused_valu_insts_per_launch")
        if n:
            f = {c: t.get(f"fused_f64_{c}_insts_per_launch", 0.0) for c in ("fma", "mul", "add", "trans")}
            out["pmc_fused_per_launch"] = {"valu_insts": n, "f64": f, "valu_not_f64_arith": n - sum(f.values()),
                                           "share_not_f64_arith": (n - sum(f.values())) / n,
                                           "lane_utilisation": t.get("fused_valu_lane_utilisation")}
    json.dump(out, open(a.out, "w"), indent=1)
    for tag, s in out["kernels"].items():
        c = s["classes"]
        print(f"{tag}: {s['total']} insts, VGPR {s.get('NumVgprs')} scratch {s.get('ScratchSize')} B | valu {s['valu']} "
              f"(f64 arith {s['f64_counted_by_pmc']}, div scaffolding {c.get('f64_div_fixup', 0)}, f64 other {c.get('f64_other', 0)}, "
              f"cndmask {c.get('cndmask', 0)}, mov {c.get('mov', 0)}, int {c.get('int', 0)}, cmp_int {c.get('cmp_int', 0)}) | "
              f"salu {c.get('salu', 0)} branch {c.get('branch', 0)} scratch {c.get('scratch', 0)} global {c.get('global', 0)} "
              f"lds {c.get('lds', 0)} waitcnt {c.get('waitcnt', 0)}")


if __name__ == "__main__":
    main()
