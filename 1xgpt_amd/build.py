"""Build libgenie_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python 1xgpt_amd/build.py            # rebuild if sources are newer than the library
    python 1xgpt_amd/build.py --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgenie_hip.so")
SOURCES = ["api.hip", "kernels_exact.hip", "kernels_bf16.hip", "kernels_gemm_pp.hip", "kernels_gemm_sm.hip", "kernels_gemm_tn.hip", "kernels_attn16.hip", "kernels_attn_dma.hip", "kernels_fused.hip", "kernels_fused_prefix.hip", "kernels_fused_f16x3.hip", "kernels_frame.hip", "kernels_conv.hip",
           "kernels_train.hip", "kernels_train16.hip", "kernels_attn_bwd16.hip", "train_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# per-source extras.  kernels_frame.hip: its launches are a few microseconds long and begin by fetching their arguments -- let the
# command processor preload them into SGPRs (same-box A/B on generate: +1.2 % at batch 1, profiles/r05z_kernarg_preload_ab.txt)
FILE_FLAGS = {"kernels_frame.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, study=None):
    """study=True (or GENIE_STUDY=1 in the environment of `python 1xgpt_amd/build.py`): compile with -DGENIE_STUDY into
    libgenie_hip_study.so -- the ablation / scheduling / reduced-precision knobs of the kernels' A/B studies are live there and
    only there (load it with GENIE_HIP_LIBRARY=<path>); the shipping library reads no environment variable on a launch path."""
    if study is None:
        study = os.environ.get("GENIE_STUDY", "0") == "1"
    if study:
        return _build(force, verbose, os.path.join(HERE, "libgenie_hip_study.so"), os.path.join(HERE, "build", "study"),
                      FLAGS + ["-DGENIE_STUDY"])
    return _build(force, verbose, LIB, os.path.join(HERE, "build"), FLAGS)


def _build(force, verbose, LIB, OBJDIR, FLAGS):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "genie_hip.h"))
    objs = []
    os.makedirs(OBJDIR, exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, "-x", "hip", "-c", s, "-o", o] + FLAGS + FILE_FLAGS.get(src, [])
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines, verbose=False):
    """An A/B variant of the SHIPPING library (same flags plus `defines`, e.g. ["-DGENIE_VAR_NO_LNOUT"]) in
    1xgpt_amd/lib_ab_<name>.so (in-tree so that it travels to the GPU box; *.so is git-ignored) -- loaded with
    GENIE_HIP_LIBRARY=<path> by `tools/gpu_run.sh ab <tag> <name> ...` for same-box comparisons."""
    return _build(False, verbose, os.path.join(HERE, f"lib_ab_{name}.so"), os.path.join(HERE, "build", "ab", name), FLAGS + list(defines))


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python 1xgpt_amd/build.py --variant <name> [-DX ...]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith(("-D", "-f", "-m"))], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
