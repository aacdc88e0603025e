#!/usr/bin/env python3
"""Static check of a built library against the pattern tools/ubench/pk_lds_repro.hip reproduces on gfx950 (round 5):
a packed fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) that selects operand halves with op_sel /
op_sel_hi from a register an LDS read wrote returns wrong values in lanes 32-63 while another wave of the SIMD issues
16-bit MFMAs - in 100 % of the launches of the reproducer, never for the same instruction on registers a VALU instruction
wrote, never without the op_sel selection, never for v_fma_f32 pairs (profiles/NOTES_r05.md section 2).

The device code of the library is disassembled (llvm-objdump) and every kernel is walked in program order with the last
writer of each VGPR: `flagged` = packed fp32 instructions with an op_sel modifier on a source register last written by a
ds_read.  Program order ignores control flow, so the walk can miss a path or see one that does not exist; it is a lint, the
stress tests (tests/test_gpu_shared_gpu.py, tests/test_gpu_jitter.py) are the proof.

    python tools/lint_pk_lds.py [remora_amd/libremora_hip.so] [--json]
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(tok):
    out = []
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint(path):
    tmp = tempfile.mkdtemp()
    try:
        lib = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
        res = {"library": path, "code_objects": len(objs), "kernels": 0, "packed_f32": 0, "packed_f32_with_op_sel": 0,
               "packed_f32_reading_lds_written_registers": 0, "flagged": []}
        for obj in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", obj], check=True, capture_output=True, text=True).stdout
            kernel, writer = None, {}
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
                if m:
                    kernel, writer = m.group(1), {}
                    res["kernels"] += 1
                    continue
                t = ln.strip().split("//")[0].strip()
                if not t or kernel is None:
                    continue
                op, _, rest = t.partition(" ")
                ops = [x.strip() for x in rest.split(",")]
                if op.startswith("v_pk_") and op.endswith("_f32"):
                    res["packed_f32"] += 1
                    # per source operand: does op_sel / op_sel_hi pick its halves differently from the default (op_sel 0, op_sel_hi 1)?
                    src_ops = [x for x in ops[1:] if "op_sel" not in x and not x.startswith(("neg_", "clamp"))]
                    sel = [False] * len(src_ops)
                    for name, dflt in (("op_sel_hi", 1), ("op_sel", 0)):
                        mm = re.search(name + r":\[([01,]+)\]", t)
                        if mm:
                            for k, bit in enumerate(mm.group(1).split(",")):
                                if k < len(sel) and int(bit) != dflt:
                                    sel[k] = True
                    lds_ops = [any(writer.get(r) == "lds" for r in vregs(x)) for x in src_ops]
                    res["packed_f32_reading_lds_written_registers"] += any(lds_ops)
                    if any(sel):
                        res["packed_f32_with_op_sel"] += 1
                        if any(a and b for a, b in zip(sel, lds_ops)):
                            res["flagged"].append({"kernel": kernel, "instruction": t})
                    for r in vregs(ops[0]):
                        writer[r] = "valu"
                    continue
                dst = vregs(ops[0]) if ops and ops[0].startswith("v") else []
                if op.startswith("ds_read") or op.startswith("ds_load"):
                    for r in dst:
                        writer[r] = "lds"
                elif op.startswith(("v_", "global_load", "buffer_load", "flat_load", "scratch_load")) and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    for r in dst:
                        writer[r] = "other"
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = lint(args[0] if args else os.path.join(root, "remora_amd", "libremora_hip.so"))
    if "--json" in sys.argv:
        print(json.dumps(out))
    else:
        print(f"{out['library']}: {out['kernels']} kernels, {out['packed_f32']} packed fp32 instructions, {out['packed_f32_with_op_sel']} with op_sel, "
              f"{out['packed_f32_reading_lds_written_registers']} on registers an LDS read wrote, {len(out['flagged'])} flagged (both)")
        by_kernel = {}
        for f in out["flagged"]:
            by_kernel.setdefault(f["kernel"], []).append(f["instruction"])
        for k, v in by_kernel.items():
            print(f"  {len(v):4d} in {k[:100]}   e.g. {v[0]}")
    sys.exit(1 if out["flagged"] else 0)
