#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures of libturbogp.so's gfx950 code objects, read from
the AMDGPU metadata notes (what the compiler settled on, not what the source hopes for).

    python tools/kernel_resources.py [--spills-only] [pattern ...]

Prints one line per kernel: name, VGPRs, AGPRs, SGPRs, VGPR / SGPR spills, LDS bytes, scratch bytes.
Exit status 1 with --spills-only when any kernel spills (the opt-in sweep arithmetics' cross-kernel instances and the
refine kernels do; the headline f32 / f64 instances and the fit's kernels do not)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "turbo_amd", "csrc")


def code_objects(tmp):
    """llvm-objdump --offloading drops one file per embedded gfx950 code object NEXT TO its input:
    work on a copy in the temporary directory, never in the tree"""
    import shutil
    so = os.path.join(tmp, "libturbogp.so")
    shutil.copy(os.path.join(CSRC, "libturbogp.so"), so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, cwd=tmp,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def kernels_of(co):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                           capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk

        def f(key, default="0"):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else default
        name = f("name", "?")
        try:
            dem = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        out.append(dict(name=dem, vgpr=int(f("vgpr_count")), agpr=int(f("agpr_count")), sgpr=int(f("sgpr_count")),
                        vspill=int(f("vgpr_spill_count")), sspill=int(f("sgpr_spill_count")),
                        lds=int(f("group_segment_fixed_size")), scratch=int(f("private_segment_fixed_size"))))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    spills_only = "--spills-only" in sys.argv
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(tmp):
            fn = os.path.basename(co).split(".hipv4")[0]
            for k in kernels_of(co):
                if args and not any(a in k["name"] for a in args):
                    continue
                if k["vspill"]:
                    bad += 1
                if spills_only and not (k["vspill"] or k["sspill"]):
                    continue
                print("%-28s %s  vgpr %d agpr %d sgpr %d  spill v%d s%d  lds %d  scratch %d" % (
                    fn, k["name"][:110], k["vgpr"], k["agpr"], k["sgpr"], k["vspill"], k["sspill"], k["lds"], k["scratch"]))
    return 1 if (spills_only and bad) else 0


if __name__ == "__main__":
    sys.exit(main())
