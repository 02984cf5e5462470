"""bench.py's in-run power sampler (timing.power_while_stepping) must never be able to break the bench line: no rocm-smi -> None; a tool whose
output has another shape -> None; the expected shape -> median / max over the samples.  Runs without a GPU (a stand-in executable)."""
import os
import stat
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _fake_smi(tmp_path, body):
    exe = tmp_path / "rocm-smi"
    exe.write_text("#!/bin/sh\n" + body)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    return str(tmp_path)


def _run(monkeypatch, path_dir, seconds=0.6):
    import bench
    monkeypatch.setenv("PATH", path_dir + os.pathsep + os.environ.get("PATH", ""))
    s = bench._PowerSampler().start()
    time.sleep(seconds)
    s.stop()
    return s.summary()


def test_expected_output_is_parsed(tmp_path, monkeypatch):
    d = _fake_smi(tmp_path, 'echo "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1315.0"\necho "GPU[0]\t\t: sclk clock level: 1: (2089Mhz)"\n')
    out = _run(monkeypatch, d)
    assert out is not None and out["package_power_w_median"] == 1315.0 and out["package_power_w_max"] == 1315.0 and out["sclk_mhz_median"] == 2089
    assert out["samples"] >= 1


def test_unexpected_output_and_failing_tool_give_none(tmp_path, monkeypatch):
    d = _fake_smi(tmp_path, 'echo "something else entirely"\n')
    assert _run(monkeypatch, d, 0.3) is None
    (tmp_path / "x").mkdir()
    d2 = _fake_smi(tmp_path / "x", "exit 3\n")
    assert _run(monkeypatch, d2, 0.3) is None
