"""Static picture of a kernel's hot loop from the device assembly (python -m flow_amd.build's device_asm): the innermost
loop that holds most instructions is taken as the sub-step loop; counts instructions by class inside it.
usage: python scripts/isa_loop_stats.py part.s kernel_substring [...]"""
import json
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            out[name] = body
        elif name is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                name = None
            else:
                body.append(ln.rstrip("\n"))
    return out


def stats(body):
    labels = {}
    ins = []
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        ins.append(t)
    # backward branches = loops [target, branch]
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] <= i:
                loops.append((labels[lab], i))
    if not loops:
        return None
    lo, hi = max(loops, key=lambda ab: ab[1] - ab[0])
    seg = ins[lo:hi + 1]

    def count(pat):
        return sum(1 for t in seg if re.match(pat, t))
    return {"loop_instructions": len(seg), "kernel_instructions": len(ins),
            "valu": count(r"v_"), "salu": count(r"s_(?!waitcnt|barrier|cbranch|branch|nop)"),
            "exec_mask_regions (s_and_saveexec / s_or_saveexec)": count(r"s_(and|or|andn2)_saveexec"),
            "branches": count(r"s_cbranch|s_branch"), "v_readlane": count(r"v_readlane"),
            "v_readfirstlane": count(r"v_readfirstlane"), "v_writelane": count(r"v_writelane"),
            "s_barrier": count(r"s_barrier"), "s_waitcnt": count(r"s_waitcnt"),
            "lds (ds_*)": count(r"ds_"), "ds_bpermute / ds_permute": count(r"ds_b?permute"),
            "dpp": sum(1 for t in seg if "dpp" in t or "row_" in t or "wave_sh" in t),
            "global / buffer memory": count(r"global_|buffer_|flat_"), "v_cndmask": count(r"v_cndmask")}


if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    for want in sys.argv[2:]:
        for name, body in ks.items():
            if want in name:
                print(json.dumps({"kernel": name, **(stats(body) or {})}))
