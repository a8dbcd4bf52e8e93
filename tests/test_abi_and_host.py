"""CPU-only: the C-ABI library loads and exports every symbol include/genie_hip.h declares; argument checks
that need no GPU; host-side logic (config, synthetic weights, sharding)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, pkg


def declared_symbols():
    src = open(os.path.join(REPO, "include", "genie_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(genie_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib_mod = pkg("_lib")
    if not os.path.exists(lib_mod.LIB_PATH):
        pkg("build").build()
    L = lib_mod.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in genie_hip.h but not exported"
        assert n in lib_mod.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(lib_mod.SIGNATURES) == names
    assert L.genie_version() == 3


def test_struct_layout_matches_header():
    """The ctypes declarations of _lib.py against the COMPILER's layout of include/genie_hip.h (genie_abi_layout): sizes of the
    four POD structs and the offsets of the fields ABI versions 2 and 3 added (fused streams, f16x3 range flags, frame streams)."""
    lib_mod = pkg("_lib")
    L = lib_mod.load()
    out = (ctypes.c_size_t * 12)()
    assert L.genie_abi_layout(out, 12) == 12
    A, LW, W = lib_mod.AttnWeights, lib_mod.LayerWeights, lib_mod.Weights
    mine = [ctypes.sizeof(lib_mod.GenieCfg), ctypes.sizeof(A), A.fused_w16.offset, A.w16_wide.offset, ctypes.sizeof(LW),
            LW.mlp_fused_w16.offset, LW.w16_wide.offset, ctypes.sizeof(W), W.out_w16_wide.offset, A.frame_w16.offset,
            LW.mlp_frame_w16.offset, W.out_frame_w16.offset]
    assert list(out) == mine, (list(out), mine)
    assert ctypes.sizeof(lib_mod.GenieCfg) == 18 * 4


def test_config_checks_without_gpu():
    lib_mod = pkg("_lib")
    L = lib_mod.load()
    C = pkg("config")
    ok = lib_mod.make_cfg(C.c138())
    assert L.genie_check_config(ok) == 0
    assert L.genie_workspace_bytes(ok, 1) > 4096 * 512 * 4 * 2
    assert L.genie_workspace_bytes(ok, 2) > L.genie_workspace_bytes(ok, 1)
    bad = lib_mod.make_cfg(C.GenieConfig(num_layers=1, num_heads=3, d_model=96 * 3, num_factored_vocabs=2))
    assert L.genie_check_config(bad) == lib_mod.E_SHAPE  # head_dim 96 unsupported
    assert b"head_dim" in L.genie_last_error()
    assert L.genie_workspace_bytes(bad, 1) == 0
    # NULL / range checks fire before any HIP call
    assert L.genie_layer_norm(0, 0, 0, 0, 4, 64, 1e-5, 0) == lib_mod.E_ARG
    assert L.genie_mask_step(0, 3, 0, 262144, 0, 0, 0, 0, 1, 256, 0) == lib_mod.E_ARG
    with pytest.raises(lib_mod.GenieHipError):
        lib_mod.check(L.genie_linear(0, 0, 0, 0, 1, 1, 16, 0, 0, 0), "genie_linear")


def test_config_roundtrip_and_derived(tmp_path):
    C = pkg("config")
    c = C.c35()
    assert c.factored_vocab_size == 512 and c.mask_token_id == 262144 and c.head_dim == 32
    assert abs(c.attn_scale - 32 ** -0.5) < 1e-12 and c.readout_mult == 1.0
    m = C.GenieConfig(num_layers=2, num_heads=8, d_model=512, use_mup=True, num_factored_vocabs=2)
    assert m.attn_scale == 8 / 64 and m.readout_mult == 0.5
    p = tmp_path / "config.json"
    c.save_pretrained(p)
    assert C.GenieConfig.from_pretrained(p) == c
    # the reference's shipped JSON loads unchanged (same field names)
    import json
    shipped = {"num_layers": 32, "num_heads": 8, "d_model": 256, "T": 16, "S": 256, "image_vocab_size": 262144,
               "use_mup": False, "num_factored_vocabs": 2, "qkv_bias": False, "proj_bias": True, "attn_drop": 0.0,
               "qk_norm": False, "mlp_ratio": 4.0, "mlp_drop": 0.0, "mlp_bias": True}
    p.write_text(json.dumps(shipped))
    assert C.GenieConfig.from_pretrained(p) == c


def test_synthetic_is_deterministic_and_complete():
    C, S = pkg("config"), pkg("synthetic")
    cfg = C.GenieConfig(num_layers=2, num_heads=2, d_model=64, T=4, S=16, num_factored_vocabs=2, qk_norm=False)
    a, b = S.make_state_dict(cfg, seed=5), S.make_state_dict(cfg, seed=5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["out_x_proj.weight"], S.make_state_dict(cfg, seed=6)["out_x_proj.weight"])
    n_params = sum(v.size for v in S.make_state_dict(C.c35(), seed=0, law="init").values())
    assert n_params == 35_218_688  # the shipped config is GENIE_35M
    assert sum(int(np.prod(s)) for _, s, _, _ in S.state_dict_spec(C.c138())) == 137_545_216
    # the module's state dict has exactly these keys
    import torch  # noqa: F401
    m = pkg("st_mask_git").STMaskGIT(cfg)
    assert sorted(m.state_dict()) == sorted(a)
    q = C.GenieConfig(num_layers=1, num_heads=2, d_model=64, T=4, S=16, num_factored_vocabs=2, qk_norm=True)
    assert sorted(pkg("st_mask_git").STMaskGIT(q).state_dict()) == sorted(S.make_state_dict(q))


def test_checkpoint_roundtrip(tmp_path):
    import torch
    C, S = pkg("config"), pkg("synthetic")
    cfg = C.GenieConfig(num_layers=1, num_heads=2, d_model=32, T=4, S=16, num_factored_vocabs=2, qk_norm=False)
    M = pkg("st_mask_git").STMaskGIT
    m = M(cfg).load_numpy_state_dict(S.make_state_dict(cfg, seed=1))
    m.save_pretrained(tmp_path)
    assert sorted(os.listdir(tmp_path)) == ["config.json", "model.safetensors"]
    m2 = M.from_pretrained(tmp_path)
    assert m2.config == cfg
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())


def test_no_cpu_fallback():
    import torch
    C = pkg("config")
    cfg = C.GenieConfig(num_layers=1, num_heads=2, d_model=32, T=4, S=16, num_factored_vocabs=2, qk_norm=False)
    m = pkg("st_mask_git").STMaskGIT(cfg)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.compute_logits(torch.zeros(1, 4, 4, 4, dtype=torch.long))
    with pytest.raises(RuntimeError, match="GPU only"):
        pkg("attention").SelfAttention(2, 32, qk_norm=False)(torch.zeros(1, 4, 32))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "1xgpt_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "genie_oracle" not in src, f


def test_shard_range_and_means():
    D = pkg("distributed")
    for n, w in [(512, 8), (10, 3), (3, 8), (64, 1)]:
        parts = [D.shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
    m = D.means_from_sums([20.0, 2.0, 3.0, 12.0, 30.0, 2.0])
    assert m == dict(loss=10.0, acc=0.25, frames=30, clips=2)


def test_header_is_plain_c_and_the_library_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: include/genie_hip.h must compile as plain C99 (no C++-isms, no torch / HIP types in the
    signatures) and a C program must link against libgenie_hip.so and call it -- what a cgo / JNI / FFI binding of the reference's
    maintainer would do.  No GPU: genie_version / genie_workspace_bytes / genie_check_config are host-only."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    lib = pkg("_lib").LIB_PATH
    if not os.path.exists(lib):
        pytest.skip("library not built")
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "genie_hip.h"\n'
                   "int main(void) {\n"
                   "    genie_cfg c; memset(&c, 0, sizeof c);\n"
                   "    c.num_layers = 2; c.num_heads = 8; c.head_dim = 64; c.d_model = 512; c.T = 16; c.S = 256; c.hidden = 2048;\n"
                   "    c.factored_vocab = 512; c.num_factored = 2; c.image_vocab_size = 262144; c.attn_scale = 0.125f; c.readout_mult = 1.0f;\n"
                   "    c.precision = GENIE_PREC_F16X3;\n"
                   "    if (genie_version() != GENIE_ABI_VERSION) return 1;\n"
                   "    if (genie_check_config(&c) != GENIE_OK) { puts(genie_last_error()); return 2; }\n"
                   "    if (genie_workspace_bytes(&c, 4) == 0 || genie_generate_workspace_bytes(&c, 4, 8) < genie_workspace_bytes(&c, 4)) return 3;\n"
                   "    c.head_dim = 48;\n"
                   "    if (genie_check_config(&c) == GENIE_OK || strlen(genie_last_error()) == 0) return 4;   /* errors are codes + text, never aborts */\n"
                   '    printf("%zu\\n", genie_prefix_cache_bytes(&c, 1));\n'
                   "    return 0;\n}\n")
    exe = tmp_path / "abi"
    inc = os.path.join(REPO, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "abi.o")], check=True)
    subprocess.run([gcc, str(tmp_path / "abi.o"), lib, f"-Wl,-rpath,{os.path.dirname(lib)}", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
