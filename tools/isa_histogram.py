"""Static per-opcode histogram of a gfx950 kernel's ISA, per basic block and per natural loop.

usage: python tools/isa_histogram.py <file.s> <kernel-symbol-regex> [--json out.json] [--blocks]

<file.s> is `hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S` output (tools/isa_render_fb.sh makes it with the product's flags).
Basic blocks are split at labels and after branches; a loop is a back edge (branch to a label at or above the branch) and its body is the
span of blocks between the target and the branch (the compositing loops are laid out contiguously: verified by the listing this prints).
Instruction classes: VALU (v_*, incl. DPP / cndmask / cmp), of which `fma_class` (v_fma / v_fmac / v_mul / v_add / v_sub / v_mac f32),
`trans` (v_exp / v_rcp / v_log / v_rsq / v_sqrt), `cmp` (v_cmp*), `cndmask`, `dpp` (any VALU with a dpp modifier), `mov`, `int/other`;
SALU (s_* except s_waitcnt / s_nop / s_branch family, listed apart), LDS (ds_*), VMEM (global_ / buffer_ / scratch_ / flat_), SMEM (s_load*)."""
import collections
import json
import re
import sys


def classify(op, line):
    if op.startswith("v_"):
        sub = "other"
        if " dpp" in line or "quad_perm" in line or "row_" in line or "wave_" in line:
            sub = "dpp"
        elif re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)_", op):
            sub = "trans"
        elif op.startswith("v_cmp") or op.startswith("v_cmpx"):
            sub = "cmp"
        elif op.startswith("v_cndmask"):
            sub = "cndmask"
        elif re.match(r"v_(fma|fmac|mac|mul|add|sub|subrev|mad|pk_fma|pk_mul|pk_add)_(f32|legacy_f32)", op):
            sub = "fma_class"
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
            sub = "mov"
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
            sub = "lane"
        elif re.match(r"v_(min|max|med3|min3|max3)_", op):
            sub = "minmax"
        return "VALU", sub
    if op.startswith("ds_"):
        return "LDS", op
    if re.match(r"(global|buffer|scratch|flat)_", op):
        return "VMEM", op.split("_")[0] + ("_store" if "store" in op else "_atomic" if "atomic" in op else "_load")
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM", op
    if op in ("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep") or op.startswith("s_waitcnt"):
        return "WAIT", op
    if op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_setpc") or op.startswith("s_call"):
        return "BRANCH", op
    if op.startswith("s_"):
        return "SALU", op
    return "OTHER", op


def parse(path, sym_re):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^(%s)\S*:" % sym_re, l))
    blocks, cur = [], {"label": "entry", "ins": []}
    for l in lines[start + 1:]:
        t = l.split(";")[0].rstrip()
        if ".end_amdhsa_kernel" in l or t.strip().startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            if cur["ins"] or cur["label"] == "entry":
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        t = t.strip()
        if not t or t.startswith(".") or t.startswith("//"):
            continue
        op = t.split()[0]
        cur["ins"].append((op, t))
        if op.startswith("s_cbranch") or op == "s_branch":
            blocks.append(cur)
            cur = {"label": cur["label"] + "+", "ins": []}
    blocks.append(cur)
    return [b for b in blocks if b["ins"]]


def summarise(ins):
    cls, sub, ops = collections.Counter(), collections.Counter(), collections.Counter()
    for op, t in ins:
        c, s = classify(op, t)
        cls[c] += 1
        ops[op + (" dpp" if c == "VALU" and s == "dpp" else "")] += 1
        if c == "VALU":
            sub[s] += 1
    return cls, sub, ops


def main():
    path, sym = sys.argv[1], sys.argv[2]
    blocks = parse(path, sym)
    index = {}
    for i, b in enumerate(blocks):
        index.setdefault(b["label"].rstrip("+"), i)
    loops = []
    for i, b in enumerate(blocks):
        op, t = b["ins"][-1]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            j = index.get(tgt)
            if j is not None and j <= i:
                loops.append((j, i))
    out = {"kernel_regex": sym, "blocks": len(blocks), "loops": []}
    tot_cls, tot_sub, _ = summarise([x for b in blocks for x in b["ins"]])
    out["whole_kernel_static"] = {"classes": dict(tot_cls), "valu_subclasses": dict(tot_sub)}
    for (j, i) in sorted(loops, key=lambda p: (p[1] - p[0], p[0])):
        ins = [x for b in blocks[j:i + 1] for x in b["ins"]]
        cls, sub, ops = summarise(ins)
        inner = [(a, c) for (a, c) in loops if j <= a and c <= i and (a, c) != (j, i)]
        blist = []
        for bb in blocks[j:i + 1]:
            c2, s2, _ = summarise(bb["ins"])
            blist.append({"label": bb["label"], "instructions": len(bb["ins"]), "classes": dict(c2), "valu_subclasses": dict(s2)})
        out["loops"].append({"head": blocks[j]["label"], "tail_block": blocks[i]["label"], "blocks": i - j + 1, "instructions": len(ins), "block_list": blist,
                             "contains_loops": [blocks[a]["label"] for a, _ in inner], "classes": dict(cls), "valu_subclasses": dict(sub),
                             "opcodes": dict(ops.most_common())})
    if "--blocks" in sys.argv:
        for i, b in enumerate(blocks):
            cls, sub, _ = summarise(b["ins"])
            print(f"{i:4d} {b['label']:16s} n={len(b['ins']):4d} {dict(cls)} {dict(sub)} -> {b['ins'][-1][1][:60]}")
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    for lp in out["loops"]:
        print(f"loop {lp['head']} .. {lp['tail_block']}: {lp['blocks']} blocks, {lp['instructions']} instr, classes {lp['classes']}, VALU {lp['valu_subclasses']}, inner {lp['contains_loops']}")


if __name__ == "__main__":
    main()
