"""The bench lines committed with the round's profiles (profiles/r05_bench.json: what `python3 bench.py` printed on an
MI355X; r05_bench_under_rocprof.json: the same command under rocprofv3) carry every field of the measurement contract,
and their numbers hang together with the committed kernel table and counter summaries."""
import json
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_follows_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_under_rocprof.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mpixels/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["metric"].split(" at ")[0] in base["metric"] or "4K PBR ShaderBall" in d["metric"]
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    # value = pixels per step / time per step
    px = d["config"]["width"] * d["config"]["height"]
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert r["algorithmic_bytes_per_launch"] == r["n_shaded"] * 36 + 6432 + 144      # 16 B + 5 texels per shaded pixel
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    assert d["parity_vs_oracle"]["bit_exact"] is True
    # the per-kernel table of the same run is committed beside it; the mean duration of k_shade's timed launches by the
    # kernel trace is what the HIP events bracket, minus what the events also see (they are recorded on the frame's
    # stream in front of the tail instantiation and behind the main one, while two other frames share the GPU)
    names = [l.split(",")[0] for l in open(os.path.join(ROOT, "profiles", "r05_bench_kernel_stats.csv")).read().splitlines()[1:]]
    assert r["kernel"] == "k_shade" and {"k_shade", "k_shade_tail", "k_shade_items", "k_raster", "k_geometry"} <= set(names)
    phases = open(os.path.join(ROOT, "profiles", "r05_bench_kernel_phases.txt")).read()
    line = [l for l in phases.splitlines() if l.startswith("k_shade ")][0]
    timed = float(line.split("mean=")[1].split("us")[0])
    alone = float(line.split("mean=")[2].split("us")[0])
    assert 0.6 < timed * 1e-3 / r["avg_kernel_ms"] < 1.05
    # the kernel alone: the trace's duration against the HIP events of the one-frame pass of the SAME run (both under the
    # profiler: the events also bracket the dispatch gaps and the profiler's interception) -- 0.82 in round 5's take
    assert 0.75 < alone * 1e-3 / r["one_frame_in_flight"]["avg_kernel_ms"] < 1.05
    # ... and against the line bench.py printed WITHOUT the profiler (the bound that pins the relationship)
    plain = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))["roofline"]
    assert 0.75 < alone * 1e-3 / plain["one_frame_in_flight"]["avg_kernel_ms"] < 1.05
    assert plain["trace_kernel_alone_ms"] == round(alone * 1e-3, 5)


def test_line_says_what_the_host_and_the_clocks_were_doing():
    """the line carries the host's side of the timed region (time inside the submit, time blocked on a
    frame slot), this GPU's clocks and power over the timed region, the warm-up's block history, and the all-cores CPU figure"""
    for name in ("r05_bench.json", "r05_bench_driver_style.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        h = d["host"]
        assert h["steps"] == d["steps"] and 0 < h["blocked_us_per_step"] < h["submit_us_per_step"] <= h["loop_us_per_step"] * 1.02
        assert h["loop_us_per_step"] <= d["ms_per_step"] * 1e3 * 1.02 and 0 <= h["blocked_steps"] <= h["steps"]
        # the GPU, not the host, sets the rate: what the host does itself per step is a fraction of the frame period
        assert h["submit_us_per_step"] - h["blocked_us_per_step"] < 0.5 * d["ms_per_step"] * 1e3
        c = d["clocks"]
        assert c["available"] and c["samples"] >= 1 and 500 < c["sclk_mhz"]["min"] <= c["sclk_mhz"]["max"] <= 2500 and c["power_w"]["mean"] > 0
        w = d["warm_up"]
        assert len(w["blocks_ms_per_step"]) >= 3 and w["frames"] == w["block_steps"] * len(w["blocks_ms_per_step"])
        if w["stopped_because"].startswith("two consecutive"):
            a, b = w["blocks_ms_per_step"][-2:]
            assert abs(a - b) <= 0.01 * a
        cb = d["cpu_baseline"]
        assert cb["all_cores"] >= cb["cores"] and cb["all_cores_frames"] >= 5 and cb["all_cores_value"] > cb["single_thread_value"]
        f2 = d["roofline"]["frames_in_flight_2"]
        assert f2["host"]["blocked_us_per_step"] > 0 and f2["ms_per_step"] > d["ms_per_step"]


def test_default_line_carries_the_1080p_figure_too():
    """north_star asks for 1080p and 4K: the default (C3) run adds C2 -- ShaderBall, one point light, 1920x1080 -- after its
    timed region, in a context of its own with that workload's four frames in flight"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    c2 = d["also"]["c2_1080p"]
    assert c2["workload"].startswith("C2:") and "1920x1080" in c2["workload"] and c2["frames_in_flight"] == 4 and c2["steps"] >= 100
    assert abs(c2["value"] - 1920 * 1080 / (c2["ms_per_step"] * 1e-3) / 1e6) / c2["value"] < 1e-3 and c2["unit"] == "Mpixels/s"
    assert c2["n_shaded"] == json.load(open(os.path.join(ROOT, "tests", "golden", "n_shaded.json")))["c2"]["n_shaded"]
    assert "also" not in json.load(open(os.path.join(ROOT, "profiles", "r05_c5_bench.json")))   # (only the headline workload's line)


def test_traffic_and_valu_come_from_summaries_of_the_same_kernel_sources():
    """roofline.traffic / roofline.valu of the line bench.py printed without the profiler are the committed counter
    summaries' figures, and those summaries say which kernel sources they were measured on"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    r = d["roofline"]
    hbm = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_hbm.json")))
    sq = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_sq.json")))
    assert len(hbm["kernel_source_sha256"]) == 64 and hbm["kernel_source_sha256"] == sq["kernel_source_sha256"]
    assert r["traffic_source"] == "profiles/r05_pmc_hbm.json" and r["traffic"] == hbm["kernels"]["k_shade"]["hbm_bytes_per_launch"]
    assert r["traffic"] <= 1.8 * r["algorithmic_bytes_per_launch"]
    v = r["valu"]
    assert v["bound"] == "valu" and v["source"] == "profiles/r05_pmc_sq.json"
    assert v["valu_instructions_per_launch"] == int(sq["kernels"]["k_shade"]["SQ_INSTS_VALU"])
    assert abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3 and sum(v["by_class"].values()) == v["valu_instructions_per_launch"]
    assert abs(v["achieved"] - v["valu_instructions_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / v["achieved"] < 1e-3
    # a summary of other sources is refused (the helper, on a made-up hash)
    import bench
    from bibim_renderer_amd import build_id
    real = build_id.kernel_source_sha256
    try:
        build_id.kernel_source_sha256 = lambda: "0" * 64
        traffic, why = bench.profiled_traffic("c3")
        assert traffic is None and "other kernel sources" in why
        assert bench.valu_roofline("c3", "k_shade", 0.1, 256)["achieved"] is None
    finally:
        build_id.kernel_source_sha256 = real


def test_bench_with_gpus_n_and_no_launcher_starts_its_own_ranks_before_touching_torch():
    """`python bench.py --gpus 8 ...` (the driver's N = 1 command form with another N): the ranks of an attempt are started as
    children of torch.distributed.run before this process has imported torch or made a HIP call, with the same arguments
    (+ the attempt's --exchange / --gather and --worker), rendezvous on 127.0.0.1; the JSON line of the attempt that
    completes is relayed and the exit code is 0."""
    import subprocess
    import sys
    code = r'''
import json, os, sys
for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BBR_BENCH_BACKEND"):
    os.environ.pop(k, None)
sys.argv = ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
import bench
seen = []
def fake_attempt(cmd, env, watch, limits, flag=None, deadline=None):
    seen.append((cmd, env, watch, flag))
    return True, '[rccl banner]\n{"metric": "m", "n_gpus": 8}\n', None, None
bench.run_attempt = fake_attempt
try:
    bench.main()
    raise AssertionError("main() returned")
except SystemExit as e:
    assert e.code == 0, e.code
assert "torch" not in sys.modules, "torch was imported before the ranks were started"
(c, env, watch, flag), = seen
assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in c and c[c.index("--nproc-per-node") + 1] == "8"
assert c[c.index("--master-addr") + 1] == "127.0.0.1"
tail = c[c.index([x for x in c if x.endswith("bench.py")][0]) + 1:]
assert tail == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--exchange", "native", "--gather", "packed", "--worker"], tail
assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and json.loads(env["BBR_BENCH_ATTEMPTS"]) == [] and watch == list(range(8)) and flag is None
print("ok")
'''
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().splitlines()[-1] == "ok" and '{"metric": "m", "n_gpus": 8}' in p.stdout, p.stderr[-2000:]


def test_attempt_ladder_falls_back_in_fresh_processes_and_reports_what_failed():
    """native hangs -> the supervisor gives the attempt up and starts torch in new processes; that attempt is told (and its
    line says) what was tried before.  Every attempt failing is exit code 1 and no line."""
    import bench
    assert bench.attempt_ladder(None, "packed", False) == [("native", "packed"), ("torch", "packed"), ("torch", "rgba32f")]
    assert bench.attempt_ladder("torch", "rgba32f", False) == [("torch", "rgba32f")]
    assert bench.attempt_ladder("peer", "rgba16f", False) == [("peer", "rgba16f"), ("torch", "rgba16f"), ("torch", "rgba32f")]
    assert bench.attempt_ladder("native", "packed", True) == [("native", "packed"), ("torch", "packed")]   # --present gathers RGBA8 either way
    assert bench._worker_argv(["--gpus", "2", "--exchange", "peer", "--gather=packed", "--verify"], "torch", "rgba32f") == \
        ["--gpus", "2", "--verify", "--exchange", "torch", "--gather", "rgba32f", "--worker"]
    import argparse
    args = argparse.Namespace(gpus=2, exchange=None, gather="packed", present=False, steps=20, warmup=5, cpu_budget=20.0)
    calls = []

    def fake(outcomes):
        def run(cmd, env, watch, limits, flag=None, deadline=None):
            calls.append((cmd[cmd.index("--exchange") + 1], cmd[cmd.index("--gather") + 1], json.loads(env["BBR_BENCH_ATTEMPTS"])))
            return outcomes[len(calls) - 1]
        return run
    real, env0 = bench.run_attempt, dict(os.environ)
    try:
        for k in ("WORLD_SIZE", "RANK", "BBR_BENCH_BACKEND"):
            os.environ.pop(k, None)
        bench.run_attempt = fake([(False, "", "no progress for 60 s while waiting for milestone 'first_frame'", None), (True, '{"ok": 1}\n', None, None)])
        assert bench.supervise(args, ["--gpus", "2"]) == 0
        assert [c[:2] for c in calls] == [("native", "packed"), ("torch", "packed")]
        assert calls[0][2] == [] and calls[1][2] == [{"exchange": "native", "gather": "packed",
                                                      "gave_up_because": "no progress for 60 s while waiting for milestone 'first_frame'"}]
        calls.clear()
        bench.run_attempt = fake([(False, "", "exit code 1 after milestone 'group'", None)] * 3)
        assert bench.supervise(args, ["--gpus", "2"]) == 1 and len(calls) == 3 and len(calls[2][2]) == 2
        # a failure that is not the exchange's (parity, shaded-pixel count: the worker says so) ends the ladder at once
        calls.clear()
        bench.run_attempt = fake([(False, "", "rank 0: GPU frame outside the 1e-4 tolerance", "rank 0: GPU frame outside the 1e-4 tolerance")] * 3)
        assert bench.supervise(args, ["--gpus", "2"]) == 1 and len(calls) == 1
    finally:
        bench.run_attempt = real
        os.environ.clear()
        os.environ.update(env0)


def test_supervisor_kills_a_stalled_attempt_and_lets_a_finished_one_through(tmp_path):
    """run_attempt on real child processes (no GPU, no torch): milestones arrive through the file named by
    BBR_BENCH_MILESTONES; a child that stops making progress is killed with its whole process group, one that reaches
    'done' is a success even if it then dawdles"""
    import bench
    import sys
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, time\n"
        "ms = os.environ['BBR_BENCH_MILESTONES']\n"
        "def mark(n):\n"
        "    open(ms, 'a').write('0 %s\\n' % n)\n"
        "mode = sys.argv[1]\n"
        "for n in ('spawned', 'imported', 'group'):\n"
        "    mark(n); time.sleep(0.05)\n"
        "if mode == 'stall':\n"
        "    time.sleep(600)\n"
        "if mode == 'crash':\n"
        "    sys.exit(5)\n"
        "if mode == 'fatal':\n"
        "    open(ms + '.fatal', 'a').write('rank 0: parity broken\\n'); os._exit(3)\n"
        "for n in ('first_frame', 'warm', 'timed'):\n"
        "    mark(n)\n"
        "print('{\"metric\": \"x\"}', flush=True)\n"
        "mark('done')\n")
    limits = lambda name: 1.5   # noqa: E731
    t0 = time.monotonic()
    ok, out, why, fatal = bench.run_attempt([sys.executable, str(child), "stall"], dict(os.environ), [0], limits)
    assert not ok and "first_frame" in why and "no progress" in why and time.monotonic() - t0 < 30 and fatal is None
    ok, out, why, fatal = bench.run_attempt([sys.executable, str(child), "crash"], dict(os.environ), [0], limits)
    assert not ok and "exit code 5" in why and "group" in why and fatal is None
    ok, out, why, fatal = bench.run_attempt([sys.executable, str(child), "fatal"], dict(os.environ), [0], limits)
    assert not ok and fatal == "rank 0: parity broken" and why == fatal
    ok, out, why, fatal = bench.run_attempt([sys.executable, str(child), "fine"], dict(os.environ), [0], limits)
    assert ok and why is None and fatal is None and out.strip() == '{"metric": "x"}'
    flag = tmp_path / "flag"
    flag.write_text("")
    ok, out, why, fatal = bench.run_attempt([sys.executable, str(child), "stall"], dict(os.environ), [0], lambda n: 60.0, str(flag))
    assert not ok and "another rank" in why
    # the run's wall-time limit ends an attempt that is still making progress slowly
    ok, out, why, fatal = bench.run_attempt([sys.executable, str(child), "stall"], dict(os.environ), [0], lambda n: 60.0, None,
                                            time.monotonic() + 1.0)
    assert not ok and "wall-time limit" in why


def test_a_supervisor_that_is_terminated_takes_its_attempt_with_it(tmp_path):
    """an attempt runs in a session of its own, so a signal to the supervisor's process group does not reach
    it.  SIGTERM to the supervisor (the driver's timeout, a launcher tearing its ranks down) must end the attempt's whole
    process group -- workers left on the GPUs inside a hung collective are exactly the case the ladder exists for."""
    import signal
    import subprocess
    import sys
    pidfile = tmp_path / "pids"
    worker = tmp_path / "worker.py"
    worker.write_text(
        "import os, subprocess, sys, time\n"
        f"open({str(pidfile)!r}, 'a').write('%d\\n' % os.getpid())\n"
        "if len(sys.argv) < 2:\n"
        "    subprocess.Popen([sys.executable, __file__, 'grandchild'])\n"     # (a rank of the attempt: same process group)
        "time.sleep(600)\n")
    sup = tmp_path / "sup.py"
    sup.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "bench._install_supervisor_handlers()\n"
        f"bench.run_attempt([sys.executable, {str(worker)!r}], dict(os.environ), [0], lambda n: 600.0)\n")
    p = subprocess.Popen([sys.executable, str(sup)], cwd=ROOT)
    t0 = time.monotonic()
    while (not pidfile.exists() or len(pidfile.read_text().split()) < 2) and time.monotonic() - t0 < 30:
        time.sleep(0.1)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2, pids
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=40) == 128 + signal.SIGTERM

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:   # a zombie still answers signal 0
            return open(f"/proc/{pid}/stat").read().split(")")[1].split()[0] != "Z"
        except OSError:
            return False
    t0 = time.monotonic()
    while any(alive(q) for q in pids) and time.monotonic() - t0 < 20:
        time.sleep(0.2)
    assert not any(alive(q) for q in pids), "the attempt's processes outlived the supervisor"


def test_ranks_agree_on_success_as_well(tmp_path, monkeypatch):
    """under a launcher every rank supervises its own worker.  If rank 0's worker reached 'done' (the line is printed) while
    rank 1's stalls in teardown, rank 1's supervisor must not walk the remaining rungs alone: it finds the run's .done flag and
    leaves with 0."""
    import argparse
    import bench
    args = argparse.Namespace(gpus=2, exchange=None, gather="packed", present=False, steps=20, warmup=5, cpu_budget=20.0)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("MASTER_PORT", "29777")
    monkeypatch.delenv("BBR_BENCH_BACKEND", raising=False)
    flag_base = os.path.join("/tmp", f"bbr_bench_{os.getppid()}_29777")
    calls = []
    # rank 0: its worker completes -> the flag appears
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(bench, "run_attempt", lambda *a, **k: (calls.append("r0") or (True, '{"ok": 1}\n', None, None)))
    monkeypatch.setattr(bench.time, "sleep", lambda s: None)
    assert bench.supervise(args, ["--gpus", "2"]) == 0 and os.path.exists(flag_base + ".done")   # (stays for the ranks still on their way)
    # rank 1: its worker stalls after the others are done; the flag is there
    try:
        monkeypatch.setenv("RANK", "1")
        monkeypatch.setattr(bench, "run_attempt", lambda *a, **k: (calls.append("r1") or (False, "", "no progress for 60 s while waiting for milestone 'done'", None)))
        assert bench.supervise(args, ["--gpus", "2"]) == 0
        assert calls == ["r0", "r1"]          # one attempt each: no walk down the ladder
    finally:
        for suffix in (".done", "_attempt0.failed"):
            try:
                os.unlink(flag_base + suffix)
            except OSError:
                pass


def _latest(pattern):
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))[-1]


def test_trace_durations_quoted_in_the_bench_line_come_from_the_committed_phases_file():
    """roofline.trace_kernel_ms / trace_frac: the kernel-trace durations of the same command under rocprofv3, so that `frac` can
    be reproduced from profiles/ -- the parser picks the newest round's C3 file, not a c2 / c5 one, and ONLY a file that was
    measured on the kernel sources of the tree (its `# kernel_source_sha256` line): otherwise the line carries no trace_*
    figures and says which file it would not quote."""
    import re
    import bench
    from bibim_renderer_amd import build_id
    newest = os.path.relpath(_latest("r??_bench_kernel_phases.txt"), ROOT)
    sha = re.search(r"^# kernel_source_sha256 ([0-9a-f]{64})", open(os.path.join(ROOT, newest)).read(), re.M)
    real = build_id.kernel_source_sha256
    try:
        if sha:   # a stamped file: it is quoted exactly when the tree's kernels are the ones it was measured on
            build_id.kernel_source_sha256 = lambda: sha.group(1)
            t, alone, src = bench.committed_trace_ms("c3")
            assert src == newest and 0.04 < alone < t < 0.2
            t2, alone2, src2 = bench.committed_trace_ms("c2")
            assert re.fullmatch(r"profiles/r\d\d_c2_bench_kernel_phases.txt", src2) and alone2 < alone
        build_id.kernel_source_sha256 = lambda: "0" * 64
        t, alone, src = bench.committed_trace_ms("c3")
        assert t is None and alone is None and "other kernel sources" in src
    finally:
        build_id.kernel_source_sha256 = real
    d = json.load(open(_latest("r??_bench.json")))
    r = d["roofline"]
    if "trace_kernel_ms" in r:
        t = r["trace_kernel_ms"]
        assert abs(r["trace_frac"] - r["algorithmic_bytes_per_launch"] / (t * 1e-3) / 1e9 / r["peak"]) < 1e-3
    # the honesty fields of rounds 3 and 4
    assert d["shaded_mpixels_per_s"] < d["value"] and abs(d["shaded_fraction_of_frame"] - r["n_shaded"] / (3840 * 2160)) < 1e-3
    assert d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["cores_available"]
    assert r["frames_in_flight_2"]["ms_per_step"] > 0 and r["single_frame_device_latency_ms"] > r["one_frame_in_flight"]["avg_kernel_ms"]
    assert d["parity_vs_literal"]["within_abs_1e-4"] is True and d["parity_vs_literal"]["bit_exact"] is False
    assert d["parity_vs_oracle"]["bit_exact"] is True


def test_clock_sampler_reads_a_cards_hwmon_files(tmp_path, monkeypatch):
    """ClockSampler: plain file reads of the hwmon files of the card with the device's PCI address -- rehearsed on a made-up
    sysfs tree (the real one only exists on a GPU box); a device without readable files reports `available: False`"""
    import glob as _glob
    import bench
    dev = tmp_path / "sys" / "devices" / "pci0000:00" / "0000:c1:00.0"
    hw = dev / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("2100000000\n")
    (hw / "freq2_input").write_text("2000000000\n")
    (hw / "power1_input").write_text("650000000\n")
    card = tmp_path / "sys" / "class" / "drm" / "card7"
    card.mkdir(parents=True)
    os.symlink(dev, card / "device")
    other = tmp_path / "sys" / "class" / "drm" / "card8"
    other.mkdir()
    os.symlink(tmp_path / "sys" / "devices", other / "device")
    real_glob = _glob.glob

    def fake_glob(pattern, *a, **k):
        if pattern == "/sys/class/drm/card*/device":
            return real_glob(str(tmp_path / "sys" / "class" / "drm" / "card*" / "device"))
        return real_glob(pattern, *a, **k)
    monkeypatch.setattr(_glob, "glob", fake_glob)
    s = bench.ClockSampler("0000:C1:00.0", period_s=0.005).start()
    t0 = time.perf_counter()
    time.sleep(0.06)
    (hw / "freq1_input").write_text("1900000000\n")
    time.sleep(0.06)
    t1 = time.perf_counter()
    s.stop()
    out = s.summary(t0, t1)
    assert out["available"] and out["samples"] >= 4 and out["source"].endswith("card7/device")
    assert out["sclk_mhz"]["min"] == 1900.0 and out["sclk_mhz"]["max"] == 2100.0 and out["mclk_mhz"]["mean"] == 2000.0
    assert out["power_w"]["mean"] == 650.0
    short = s.summary(t1 + 10.0, t1 + 10.001)   # a window without a sample: the nearest ones on either side
    assert short["available"] and short["samples"] >= 1
    none = bench.ClockSampler("0000:ff:00.0")
    assert none.summary(0.0, 1.0)["available"] is False
    none.stop()


def test_cpu_quota_reads_the_cgroup(tmp_path, monkeypatch):
    import builtins
    import bench
    real_open = builtins.open

    def fake(files):
        def _open(name, *a, **k):
            if str(name) in files:
                if files[str(name)] is None:
                    raise OSError(name)
                return real_open(files[str(name)], *a, **k)
            return real_open(name, *a, **k)
        return _open
    v2 = tmp_path / "cpu.max"
    v2.write_text("1600000 100000\n")
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": str(v2)}))
    assert bench.cpu_quota() == 16.0
    v2.write_text("max 100000\n")
    assert bench.cpu_quota() is None
    q, per = tmp_path / "q", tmp_path / "p"
    q.write_text("250000\n"); per.write_text("100000\n")
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": str(q),
                                                "/sys/fs/cgroup/cpu/cpu.cfs_period_us": str(per)}))
    assert bench.cpu_quota() == 2.5
