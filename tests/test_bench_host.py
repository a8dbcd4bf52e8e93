"""Host-side pieces of bench.py that run without a GPU: the board power / clock sampler's window logic and the FLOP accounting
of a pass (the numbers the bench line's `roofline` object is built from)."""
import importlib.util
import json
import os
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # __name__ != "__main__": nothing runs
    return mod


def test_board_sampler_window(tmp_path):
    bench = load_bench()
    s = bench.BoardSampler.__new__(bench.BoardSampler)  # no helper process: feed the file it would have written
    s.proc = type("P", (), {"terminate": lambda self: None, "wait": lambda self, timeout=None: 0})()
    s.path = str(tmp_path / "board.jsonl")
    t = 1000.0
    rows = [(t - 1.0, 300.0, 150.0), (t + 0.1, 1390.0, 1600.0), (t + 0.6, 1400.0, 1500.0), (t + 5.0, 280.0, 120.0)]
    # the busy GPU is rocm-smi's card3 (a one-GPU lease on a multi-GPU box: torch calls it cuda:0), card0 idles
    def write():
        with open(s.path, "w") as f:
            f.write(json.dumps({"bus": {"card0": "0000:05:00.0", "card3": "0000:85:00.0"}}) + "\n")
            for ts, p, clk in rows:
                f.write(json.dumps({"t": ts, "cards": {"card3": (p, clk), "card0": (1.0, 2.0)}}) + "\n")
            f.write("not json\n")
    write()
    out = s.window(t, t + 1.0, "0000:85:00.0")            # matched by PCI bus id
    assert out["samples"] == 2 and out["card"] == "card3" and "pci" in out["card_matched_by"]
    assert abs(out["socket_power_w_avg"] - 1395.0) < 1e-9 and out["socket_power_w_max"] == 1400.0
    assert abs(out["sclk_mhz_avg"] - 1550.0) < 1e-9 and out["sclk_mhz_min"] == 1500.0
    assert not os.path.exists(s.path)
    write()
    out = s.window(t, t + 1.0, None)                      # no bus id: the card that draws the power
    assert out["card"] == "card3" and "power" in out["card_matched_by"] and out["samples"] == 2
    write()
    out = s.window(t, t + 1.0, "0000:05:00.0")            # the idle card, if that IS the device
    assert out["card"] == "card0" and out["socket_power_w_max"] == 1.0


def test_board_sampler_without_samples(tmp_path):
    bench = load_bench()
    s = bench.BoardSampler.__new__(bench.BoardSampler)
    s.proc = None
    s.path = str(tmp_path / "none.jsonl")
    assert s.window(0.0, time.time(), None) is None


def test_pass_flops_counts_every_linear_and_attention_product():
    bench = load_bench()
    import importlib
    cfgmod = importlib.import_module("1xgpt_amd.config")
    cfg = cfgmod.GenieConfig(num_layers=2, num_heads=4, d_model=128, T=4, S=16, num_factored_vocabs=2)
    d, hid, S, T, L = cfg.d_model, cfg.d_model * 4, cfg.S, cfg.T, cfg.num_layers
    V = cfg.factored_vocab_size * cfg.num_factored_vocabs
    rows = T * S
    linear = 2.0 * rows * (L * (3 * d * d + d * d + 3 * d * d + d * d + 2 * d * hid) + d * V)
    got = bench.pass_flops(cfg, T)
    assert got >= linear                       # attention products on top of the Linear layers
    assert got <= linear * 1.5


def test_rank_supervisor_restarts_a_stalled_first_process_once(tmp_path):
    """bench.py's per-rank parent (multi-rank runs): a child that ends with HIP_INIT_STALL_RC (its first GPU touch never
    returned) is replaced by ONE fresh process; any other exit code is passed through; a second stall is final."""
    import sys
    import pytest
    bench = load_bench()
    marker = tmp_path / "attempts"
    script = ("import os,sys; p=sys.argv[1]; n=int(open(p).read()) if os.path.exists(p) else 0; open(p,'w').write(str(n+1)); "
              "sys.exit(int(sys.argv[2 + min(n, len(sys.argv) - 3)]))")
    for codes, want_rc, want_attempts in ((["17", "0"], 0, 2), (["17", "17"], 17, 2), (["0"], 0, 1), (["5"], 5, 1)):
        if marker.exists():
            marker.unlink()
        with pytest.raises(SystemExit) as e:
            bench._supervise_rank([sys.executable, "-c", script, str(marker)] + codes)
        assert e.value.code == want_rc, (codes, e.value.code)
        assert int(marker.read_text()) == want_attempts, codes


def test_device_init_watchdog_exits_with_the_stall_code():
    """1xgpt_amd.distributed.init_device: a first GPU touch that does not return ends the process with HIP_INIT_STALL_RC (and a
    stack dump) instead of hanging the peers; exercised on CPU by making the device query sleep."""
    import subprocess
    import sys
    code = ("import importlib, sys, time, torch; sys.path.insert(0, %r); D = importlib.import_module('1xgpt_amd.distributed'); "
            "torch.cuda.is_available = lambda: time.sleep(60); D.init_device(0, 1)" % REPO)
    env = dict(os.environ, GENIE_HIP_INIT_TIMEOUT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 17, (r.returncode, r.stderr[-500:])
    assert "GPU initialisation did not return" in r.stderr
