#!/usr/bin/env python3
"""Static check of the built gfx950 code: in every kernel that stages operands global -> LDS directly
(global_load_lds_*), no wave may reach an s_barrier with LDS reads still pending -- the buffer it read is the one the
next trip's DMA overwrites.  (Round 4: the compiler had sunk that wait below a raw __builtin_amdgcn_s_barrier() in
gemm64_glds_kernel; beside the background stream one fit in ten at N = 5000 came out wrong -- tools/repeat_fit.py.)

The walk is over the CONTROL-FLOW GRAPH, backwards from each s_barrier (round 5; round 4 walked the listing linearly
and could not see a read pending across a loop back-edge or a branch target): the predecessors of an instruction
are the one before it (unless that is an unconditional branch or the end of the program) and every branch that
names its label.  A path ends at an `s_waitcnt ... lgkmcnt(0)` (nothing older is pending) or at another s_barrier
(checked on its own); a path that reaches a ds_read / ds_load first is an offence.

    python tools/check_lds_dma_barriers.py          exit status 1 and one line per offending barrier
The shipped library contains no known-racy loop any more; to prove the checker is not blind it also compiles
tools/microbench/lds_race_demo.hip (the pre-fix k-loop beside the shipped one, fit_kernels.hip's flags) and expects the
first to offend and the second to pass."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import CSRC, LLVM, ROOT, code_objects   # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernels(co):
    """(demangled name, [item, ...]) per function of a code object; an item is ('L', label) or ('I', text)"""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", "--symbolize-operands", co],
                         check=True, capture_output=True, text=True).stdout
    name, body = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m and not re.fullmatch(r"L\d+", m.group(1)):
            if name is not None:
                yield name, body
            name, body = m.group(1), []
        elif m and name is not None:
            body.append(("L", m.group(1)))
        elif name is not None and line.startswith("\t"):
            body.append(("I", line.strip().split("//")[0].strip()))
    if name is not None:
        yield name, body


def cfg(body):
    """instructions, and for each its list of predecessor indices"""
    ins, label_at = [], {}
    for kind, text in body:
        if kind == "L":
            label_at[text] = len(ins)          # the next instruction
        else:
            ins.append(text)
    preds = [[] for _ in ins]
    for i, t in enumerate(ins):
        op = t.split()[0]
        falls = not (op in ("s_branch", "s_endpgm", "s_setpc_b64") or op.startswith("s_endpgm"))
        if falls and i + 1 < len(ins):
            preds[i + 1].append(i)
        if op == "s_branch" or op.startswith("s_cbranch"):
            tgt = t.split()[-1]
            if tgt in label_at and label_at[tgt] < len(ins):
                preds[label_at[tgt]].append(i)
    return ins, preds


def drains_lds(t):
    return t.startswith("s_waitcnt") and ("lgkmcnt(0)" in t or t.strip() == "s_waitcnt 0")


def offending_barriers(body):
    ins, preds = cfg(body)
    bad = []
    for i, t in enumerate(ins):
        if not t.startswith("s_barrier"):
            continue
        seen, stack, hit = set(), list(preds[i]), None
        while stack and hit is None:
            j = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            p = ins[j]
            if drains_lds(p) or p.startswith("s_barrier"):
                continue
            if p.startswith("ds_read") or p.startswith("ds_load"):
                hit = (i, j, p)
                break
            stack.extend(preds[j])
        if hit:
            bad.append(hit)
    return bad


def check_object(co, only=None):
    """[(name, bad)] for the direct-to-LDS kernels of one code object"""
    out = []
    for name, body in kernels(co):
        if not any(k == "I" and t.startswith("global_load_lds") for k, t in body):
            continue
        if only and not re.search(only, name):
            continue
        out.append((name, offending_barriers(body)))
    return out


def build_demo(tmp):
    co = os.path.join(tmp, "lds_race_demo.co")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", "-DTGP_DEBUG_KERNELS", "-I", CSRC, "-c",
                    os.path.join(ROOT, "tools", "microbench", "lds_race_demo.hip"), "-o", co],
                   check=True, capture_output=True, text=True)
    return co


def main():
    failures, checked = 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(tmp):
            for name, bad in check_object(co):
                checked += 1
                for (i, j, p) in bad:
                    print("%s: s_barrier at instruction %d reachable with `%s` (instruction %d) pending"
                          % (name.split("(")[0], i, p.split()[0], j))
                failures += 1 if bad else 0
        # the checker against the bug it was written for
        demo = dict((re.search(r"gemm64_glds_kernel<[^>]*>", n).group(0), bad) for n, bad in check_object(build_demo(tmp), r"gemm64_glds_kernel<"))
        old = [b for n, b in demo.items() if n.endswith(", 5>")]
        new = [b for n, b in demo.items() if n.endswith(", 0>")]
        demo_ok = len(old) == 1 and len(new) == 1 and bool(old[0]) and not new[0]
        print("demonstration object: the pre-fix k-loop %s, the shipped k-loop %s -> the checker is %s"
              % ("offends" if old and old[0] else "PASSES", "passes" if new and not new[0] else "OFFENDS",
                 "not blind" if demo_ok else "BLIND"))
    print("%d kernels with direct-to-LDS staging checked in libturbogp.so, %d offending" % (checked, failures))
    sys.exit(1 if (failures or not demo_ok) else 0)


if __name__ == "__main__":
    main()
