#!/usr/bin/env python3
"""Per-kernel audit of the gfx950 code objects in libgenie_hip.so for the waits the compiler inserts around LDS-DMA
("compiler waits"): for every kernel, the number of LDS-DMA instructions (`buffer_load … lds`,
`global_load_lds_*`), of full drains `s_waitcnt vmcnt(0)`, of counted waits, of barriers, the scratch bytes and the VGPR count.
A full drain inside a kernel that keeps DMA in flight serialises the stream it was meant to overlap; the hand-written ones are
one per K-tile (or fewer), anything beyond that is the compiler's (TBAA of HIP's float4 struct, an ordinary global load beside
the DMA, `__syncthreads()`).

    python tools/isa_audit.py [lib.so] [--kernel SUBSTR] [--dump SUBSTR]   (no GPU needed)
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, tmp):
    """Every gfx950 code object of the library: .hip_fatbin holds one offload bundle per translation unit, back to back."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    outs = []
    for n, s0 in enumerate(starts):
        piece = os.path.join(tmp, f"bundle{n}.bin")
        open(piece, "wb").write(blob[s0:starts[n + 1] if n + 1 < len(starts) else len(blob)])
        out = os.path.join(tmp, f"dev{n}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out):
            outs.append(out)
    if not outs:
        raise SystemExit("no gfx950 code object found in " + lib)
    return outs


def audit(lib, kernel=None, dump=None, all_kernels=False):
    """[(demangled kernel name, #LDS-DMA, #vmcnt(0), #vmcnt(n), #barriers, #mfma, scratch bytes, vgprs)] for the library's kernels."""
    with tempfile.TemporaryDirectory() as tmp:
        dis, meta = "", ""
        for co in code_objects(lib, tmp):
            dis += "\n" + subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            meta += "\n" + subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    regs, cur = {}, {}
    for line in meta.splitlines():
        s = line.strip()
        if s.startswith("- "):
            s = s[2:]
        for key in (".name:", ".private_segment_fixed_size:", ".vgpr_count:"):
            if s.startswith(key):
                cur[key] = s.split(":", 1)[1].strip()
        if s.startswith(".wavefront_size") and len(cur) == 3:
            regs[cur[".name:"]] = (int(cur[".private_segment_fixed_size:"]), int(cur[".vgpr_count:"]))
            cur = {}
    kernels = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:\n)", dis)
    names = subprocess.run(["c++filt"], input="\n".join(re.findall(r"<([^>]+)>:", dis)), capture_output=True, text=True).stdout.split("\n")
    rows = []
    i = 0
    for k in kernels:
        m = re.match(r"[0-9a-f]+ <([^>]+)>:\n", k)
        if not m:
            continue
        mangled = m.group(1)
        dem = names[i] if i < len(names) else mangled
        i += 1
        if kernel and kernel not in dem:
            continue
        if dump and dump in dem:
            print(f"==== {dem}\n{k}")
        dma = len(re.findall(r"buffer_load_dword\w* .* lds|global_load_lds_\w+", k))
        if not dma and not all_kernels:
            continue
        full = len(re.findall(r"s_waitcnt[^\n]*vmcnt\(0\)", k))
        counted = len(re.findall(r"s_waitcnt[^\n]*vmcnt\((?!0\))\d+\)", k))
        bars = len(re.findall(r"s_barrier", k))
        mfma = len(re.findall(r"v_mfma_", k))
        scratch, vg = regs.get(mangled, (-1, -1))
        rows.append((re.sub(r"\(.*", "", dem), dma, full, counted, bars, mfma, scratch, vg))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(REPO, "1xgpt_amd", "libgenie_hip.so"))
    ap.add_argument("--kernel", default=None, help="only kernels whose demangled name contains this")
    ap.add_argument("--dump", default=None, help="print the disassembly of kernels whose name contains this")
    ap.add_argument("--all", action="store_true", help="also kernels without LDS-DMA")
    a = ap.parse_args()
    rows = audit(a.lib, a.kernel, a.dump, a.all)
    print(f"# {a.lib}\n# {'kernel':86s} {'DMA':>4s} {'vmcnt(0)':>8s} {'vmcnt(n)':>8s} {'barrier':>7s} {'mfma':>5s} {'scratch':>7s} {'vgpr':>5s}")
    for r in rows:
        print(f"{r[0][:86]:88s} {r[1]:4d} {r[2]:8d} {r[3]:8d} {r[4]:7d} {r[5]:5d} {r[6]:7d} {r[7]:5d}")


if __name__ == "__main__":
    sys.exit(main())
