#!/usr/bin/env python3
"""Static check of the built gfx950 code: in every kernel that stages operands global -> LDS directly
(global_load_lds_*), no wave may reach an s_barrier with LDS reads still pending -- the buffer it read is the one the
next trip's DMA overwrites.  Walking back from each s_barrier, an `s_waitcnt ... lgkmcnt(0)` has to come before any
ds_read / ds_load.  (Round 4: the compiler had sunk that wait below a raw __builtin_amdgcn_s_barrier() in
gemm64_glds_kernel; beside the background stream one fit in ten at N = 5000 came out wrong -- tools/repeat_fit.py.)

    python tools/check_lds_dma_barriers.py          exit status 1 and one line per offending barrier
The instantiation kept to demonstrate the old loop (gemm64_glds_kernel<.., 5>, TGP_GEMM64=round4-war) is expected to
offend and is reported separately."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, code_objects   # noqa: E402


def kernels(co):
    """(demangled name, [instruction text, ...]) per function of a code object"""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", co], check=True,
                         capture_output=True, text=True).stdout
    name, body = None, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if name is not None:
                yield name, body
            name, body = m.group(1), []
        elif name is not None and line.startswith("\t"):
            body.append(line.strip().split("//")[0].strip())
    if name is not None:
        yield name, body


def offending_barriers(body):
    bad = []
    for i, ins in enumerate(body):
        if not ins.startswith("s_barrier"):
            continue
        for j in range(i - 1, -1, -1):
            p = body[j]
            if p.startswith("s_waitcnt") and ("lgkmcnt(0)" in p or p.strip() == "s_waitcnt 0"):
                break
            if p.startswith("ds_read") or p.startswith("ds_load"):
                bad.append((i, j, p))
                break
            if p.startswith("s_barrier"):            # nothing read since the previous barrier
                break
    return bad


def main():
    failures, expected, checked = 0, 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(tmp):
            for name, body in kernels(co):
                if not any(ins.startswith("global_load_lds") for ins in body):
                    continue
                checked += 1
                bad = offending_barriers(body)
                demo = re.search(r"gemm64_glds_kernel<[^>]*, 5>", name) is not None
                for (i, j, p) in bad:
                    print("%s%s: s_barrier at instruction %d passed with `%s` (instruction %d) pending"
                          % ("(expected, the old loop) " if demo else "", name.split("(")[0], i, p.split()[0], j))
                if bad and demo:
                    expected += 1
                elif bad:
                    failures += 1
    print("%d kernels with direct-to-LDS staging checked, %d offending, %d demonstration kernels offending as expected"
          % (checked, failures, expected))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
