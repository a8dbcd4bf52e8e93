"""Properties of the COMPILED gfx950 code that the measured rates depend on ("compiler-inserted waits"), checked on the
built library without a GPU (tools/isa_audit.py disassembles the code objects in libgenie_hip.so):

* the chip-filling kernels of the benchmarked path keep nothing in scratch -- a scratch reload is an `s_waitcnt vmcnt` behind the
  LDS-DMA prefetch of the next tile and behind every earlier store;
* their only full drain `s_waitcnt vmcnt(0)` is the hand-written one (per K-tile, or one in the whole persistent kernel): the
  compiler adds more when an LDS access goes through HIP's float4 struct, when an ordinary load is issued beside LDS-DMA, when an
  epilogue is written element by element.
A regression here does not change any result; it costs 5-40 % of a kernel silently."""
import importlib.util
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def rows():
    if not (os.path.exists(f"{LLVM}/llvm-objdump") and os.path.exists(f"{LLVM}/clang-offload-bundler")):
        pytest.skip("no ROCm LLVM tools")
    pkg = importlib.import_module("1xgpt_amd._lib")
    if not os.path.exists(pkg.LIB_PATH):
        importlib.import_module("1xgpt_amd.build").build()
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(REPO, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {r[0]: r for r in mod.audit(pkg.LIB_PATH, all_kernels=True)}
    assert len(out) > 100, "the audit found too few kernels: the code-object extraction is broken"
    return out


def pick(rows, sub):
    got = {k: v for k, v in rows.items() if sub in k}
    assert got, f"no kernel named like {sub!r} in the library"
    return got


def test_phase_scheduled_gemm_flavours_have_no_scratch_and_one_drain(rows):
    # every compile-time epilogue flavour of the 256x256 kernel (the run-time-flag instantiation `-1` is known to spill and is
    # not on the benchmarked path)
    for name, (_, dma, full, _, _, mfma, scratch, vgpr) in pick(rows, "gemm16_pp_kernel<").items():
        if name.rstrip().endswith(", -1>"):
            continue
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane"
        # the qk-norm flavours (epilogue flag 128) load gamma | beta with plain loads at kernel START, before the first LDS-DMA is
        # requested: one more full drain, outside the persistent loop
        qknorm = (int(name.rstrip().rstrip(">").split(",")[-1]) & 128) != 0
        assert full <= (2 if qknorm else 1), f"{name}: {full} x s_waitcnt vmcnt(0) (1 is hand-written)"
        assert dma >= 40 and mfma >= 128 and vgpr <= 256


def test_exact_gemm_interior_path(rows):
    # no scratch; the GELU flavour (no loads in its epilogue besides the bias) has only the K-tile drains plus the edge path's
    for name, (_, dma, full, _, _, mfma, scratch, _) in pick(rows, "gemm_f32_dma_kernel<").items():
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane"
        assert dma in (16, 32) and mfma in (96, 192)
    for name, r in pick(rows, "gemm_f32_dma_kernel<16, true, false, false>").items():
        assert r[2] <= 6, f"{name}: {r[2]} x vmcnt(0)"


def test_attention_and_conv_kernels_have_no_scratch(rows):
    for sub in ("attn_spatial_dma_kernel<", "conv3x3_slab_kernel<", "conv3x3_igemm_kernel<", "wgrad16_tn_kernel",
                "attn_temporal_f32_mfma_kernel<", "attn_temporal_prefix_f32_mfma_kernel<", "layer_norm_fast_kernel<"):
        for name, r in pick(rows, sub).items():
            assert r[6] == 0, f"{name}: {r[6]} bytes of scratch per lane"
    # the slab kernel's compiler-inserted waits were ~25 before its epilogue was restructured
    for name, r in pick(rows, "conv3x3_slab_kernel<").items():
        assert r[2] <= 8, f"{name}: {r[2]} x vmcnt(0)"
    for name, r in pick(rows, "attn_spatial_dma_kernel<").items():
        assert r[2] <= 10, f"{name}: {r[2]} x vmcnt(0) (one per phase of the dynamic wait switch + prologue)"


def test_frame_kernels_have_no_scratch(rows):
    # csrc/kernels_frame.hip (the one-frame passes of generate): every operand request of a launch is in flight at once, so the
    # register-direct kernels run near 200 registers -- a spill would put a scratch round trip into a launch that is one round trip long
    n = 0
    for sub in ("gemm16_fr_kernel<", "gemm16_frm_kernel<", "attn_spatial_fr_kernel", "attn_temporal_fr_kernel", "ln_fr_kernel<"):
        for name, r in pick(rows, sub).items():
            assert r[6] == 0, f"{name}: {r[6]} bytes of scratch per lane"
            n += 1
    assert n >= 20, n


def test_fused_subblock_kernels_have_no_scratch(rows):
    # kernels_fused.hip sits at the 256-register edge (2 workgroups per CU): an innocent-looking edit makes the allocator spill,
    # and a spill in the region loop once cost 2x on the whole forward.  Main loops carry only the counted / region waits.
    for sub in ("temporal_fused_bf16_kernel<", "temporal_prefix_fused_bf16_kernel<", "temporal_qkv_attn_f16x3_kernel<", "mlp_fused_bf16_kernel<"):
        for name, r in pick(rows, sub).items():
            assert r[6] == 0, f"{name}: {r[6]} bytes of scratch per lane"
            assert r[7] <= 256 and r[5] >= 64, (name, r)
    # the spatial kernel (1 workgroup per CU, 254 registers): no scratch since round 5 -- its last spills were two lane-id-derived
    # epilogue offsets, reloaded once per sequence behind an s_waitcnt vmcnt(0) that drained the next head's LDS-DMA
    for name, r in pick(rows, "spatial_attn_proj_bf16_kernel").items():
        assert r[6] == 0, f"{name}: {r[6]} bytes of scratch per lane"
        assert r[7] <= 256 and r[5] >= 48, (name, r)
