#!/usr/bin/env python3
"""Memory / matrix / barrier instruction ORDER of the kernels of one .hip source (run-length compressed), to check that a
latency-bound kernel issues all its loads before its first wait.   python tools/isa_seq.py SRC.hip SUBSTR [SUBSTR..]"""
import re, subprocess, sys
src, subs = sys.argv[1], sys.argv[2:]
asm = "/tmp/isa_seq.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "-S", src, "-o", asm, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                "-ffp-contract=off", "--cuda-device-only"], check=True, capture_output=True)
text = open(asm).read()
parts = re.split(r"\n\t\.globl\t", text)
pat = re.compile(r"(global_load_\w+|global_store_\w+|s_waitcnt|v_mfma\w*|s_barrier|ds_read\w*|ds_write\w*|buffer_\w+|scratch_\w+|s_load_\w+|ds_bpermute\w*|s_endpgm)\b(.*)")
for p in parts[1:]:
    mangled = p.split("\n", 1)[0].strip()
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    if subs and not any(s in name for s in subs):
        continue
    body = p.split(mangled + ":", 1)[-1].split(".Lfunc_end", 1)[0]
    seq = []
    for l in body.split("\n"):
        m = pat.match(l.strip())
        if m:
            t = m.group(1)
            if t == "s_waitcnt":
                t = "W[" + m.group(2).strip().replace("vmcnt", "vm").replace("lgkmcnt", "lgkm") + "]"
            seq.append(t.replace("global_load_dwordx4", "GL4").replace("global_load_dwordx2", "GL2").replace("v_mfma_f32_32x32x16_f16", "MFMA")
                       .replace("global_store_dwordx4", "GS4").replace("ds_read_b128", "DR128"))
    out, prev, cnt = [], None, 0
    for t in seq + [None]:
        if t == prev:
            cnt += 1
        else:
            if prev:
                out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
            prev, cnt = t, 1
    print(name[:110]); print("   " + " ".join(out)); print()
