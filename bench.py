#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the forward PBR path (BASELINE.json metric) on N MI355X GPUs of one node.

A step = one frame of the workload through the C++ Scene/Camera/drawFrame shim and the C ABI:
updateScene + uniform fill + bbr_begin_frame/draw/end_frame -> k_geometry + k_raster + k_shade on the GPU, plus, for
N > 1, the RCCL all-gather of the per-rank framebuffer shards over xGMI and the un-interleave kernel (by default inside the
library, bbr_allgather_frame, on the frame's own stream).  `python bench.py --gpus N` without a launcher starts its own N
ranks (torch.distributed.run as a child process, before this process touches torch or the GPU).
Inputs (mesh, textures) are resident in HBM before the timed region; per-frame instance matrices and
uniform blocks (8 KB) are the only host->device traffic, as in the reference's render loop.

N = 1 workload: C3 = ShaderBall x16 + plane, 4 point lights, GGX PBR + normal map, 3840x2160 (the
configuration BASELINE.json's metric "4K PBR ShaderBall" and its >=100 Mpixels/s target are quoted on).
N > 1: the same frame (C4) split into interleaved screen bands => "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
XGMI_LINK_GBS_PER_DIRECTION = 76.8  # one direction of one xGMI link (153.6 GB/s both ways; seven links per MI355X)


def _committed_summary(pattern, workload, kernel):
    """newest profiles/<pattern> of this workload that holds `kernel` AND was measured on the kernels of this tree
    (kernel_source_sha256, bibim_renderer_amd/build_id.py).  Returns (entry, path, None) or (None, path_or_None, why)."""
    import glob
    from bibim_renderer_amd.build_id import kernel_source_sha256
    mine, stale = kernel_source_sha256(), None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") != workload or kernel not in d.get("kernels", {}):
            continue
        if d.get("kernel_source_sha256") == mine:
            return d["kernels"][kernel], os.path.relpath(f, ROOT), None
        stale = stale or os.path.relpath(f, ROOT)
    return None, stale, ("measured on other kernel sources than this tree's" if stale else "no summary of this workload committed")


def committed_trace_ms(workload, kernel="k_shade"):
    """mean duration of `kernel`'s timed launches by the kernel trace of the same command under rocprofv3
    (profiles/*bench_kernel_phases.txt, newest round first): what `roofline.frac` can be reproduced from.  Returns
    (timed_region_ms, alone_ms, file) or (None, None, why).  Only a file that was measured on the kernel sources of THIS tree
    counts (its `# kernel_source_sha256` line, written by tools/profile_summary.py): durations of other kernels beside this
    run's bytes would be a mixed figure."""
    import glob
    import re
    from bibim_renderer_amd.build_id import kernel_source_sha256
    mine, stale = kernel_source_sha256(), None
    pat = "*_bench_kernel_phases.txt" if workload == "c3" else f"*_{workload}_bench_kernel_phases.txt"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), reverse=True):
        if workload == "c3" and re.search(r"_c\d_bench_kernel_phases", f):
            continue
        text = open(f).read()
        sha = re.search(r"^# kernel_source_sha256 ([0-9a-f]{64})", text, re.M)
        if not sha or sha.group(1) != mine:
            stale = stale or os.path.relpath(f, ROOT)
            continue
        for line in text.splitlines():
            if line.split() and line.split()[0] == kernel:
                m = re.findall(r"mean=\s*([0-9.]+) us", line)
                if len(m) >= 2:
                    return float(m[0]) * 1e-3, float(m[1]) * 1e-3, os.path.relpath(f, ROOT)
    return None, None, (f"{stale}: measured on other kernel sources than this tree's" if stale else None)


def committed_texel_lines(workload):
    """distinct 128-byte lines of the packed material a frame of this workload touches (tools/texel_lines.py, CPU, committed as
    profiles/*_texel_lines.txt): the compulsory texel reads -- the lower bound of what the fetch counters can be DRAM traffic"""
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_texel_lines.txt")), reverse=True):
        txt = open(f).read()
        m = re.search(rf"^{workload.upper()}:.*?distinct 128-byte lines touched:\s+(\d+) = ([0-9.]+) MB", txt, re.S | re.M)
        if m:
            return int(m.group(1)) * 128, os.path.relpath(f, ROOT)
    return None, None


def profiled_traffic(workload, kernel="k_shade"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/*_pmc_hbm.json, written by
    tools/profile_summary.py: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  (None, "<file>: stale ...") when the summary belongs to another build."""
    e, path, why = _committed_summary("*_pmc_hbm.json", workload, kernel)
    if e is None:
        return None, (f"{path}: {why}" if path else why)
    return int(e["hbm_bytes_per_launch"]), path


# vector-ALU issue: one wave64 instruction per SIMD every 2 cycles is what plain fp32 / integer instructions reach with two
# or more waves on a SIMD (profiles/r02_issue_rate.txt: 2.05-2.4 cycles; 64 lanes x 2 flop / 2 cycles x 4 SIMDs x 256 CUs x
# 2.4 GHz = the 157 TFLOP/s fp32 vector peak of the data sheet)
VALU_ISSUE_CYCLES = 2.0
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector (non-matrix) peak
SHADER_CLOCK_GHZ = 2.4


def valu_roofline(workload, kernel, avg_kernel_ms, n_cus):
    """Second roofline of the dominant kernel: executed vector-ALU wave-instructions per launch (SQ_INSTS_VALU and its
    classes, profiles/*_pmc_sq.json) against the issue peak, at the kernel duration measured live in this run."""
    e, path, why = _committed_summary("*_pmc_sq.json", workload, kernel)
    peak = n_cus * 4 * SHADER_CLOCK_GHZ / VALU_ISSUE_CYCLES   # G wave-instructions / s
    out = {"bound": "valu", "kernel": kernel, "peak": round(peak, 1), "unit": "G wave-instr/s",
           "peak_is": f"{n_cus} CUs x 4 SIMDs x {SHADER_CLOCK_GHZ} GHz / {VALU_ISSUE_CYCLES} cycles per wave64 instruction",
           "achieved": None, "frac": None, "source": path if e is not None else (f"{path}: {why}" if path else why)}
    if e is None or "SQ_INSTS_VALU" not in e or avg_kernel_ms <= 0:
        return out
    n = float(e["SQ_INSTS_VALU"])
    achieved = n / (avg_kernel_ms * 1e-3) / 1e9
    classes = {k[len("SQ_INSTS_VALU_"):].lower(): int(v) for k, v in e.items() if k.startswith("SQ_INSTS_VALU_")}
    classes["other (compare, select, min/max, move, lane ops)"] = int(n - sum(classes.values()))
    # the same launch as FP32 flop/s against the data sheet's vector peak (fma = 2 flop, mul / add = 1, 64 lanes)
    flop = 64.0 * (2.0 * float(e.get("SQ_INSTS_VALU_FMA_F32", 0)) + float(e.get("SQ_INSTS_VALU_MUL_F32", 0)) +
                   float(e.get("SQ_INSTS_VALU_ADD_F32", 0)))
    tflops = flop / (avg_kernel_ms * 1e-3) / 1e12
    out.update({"fp32_tflops": round(tflops, 2), "fp32_peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                "fp32_frac": round(tflops / FP32_VECTOR_PEAK_TFLOPS, 4), "fp32_flop_per_launch": int(flop)})
    out.update({"achieved": round(achieved, 1), "frac": round(achieved / peak, 4), "valu_instructions_per_launch": int(n),
                "by_class": classes, "issue_floor_ms": round(n / (peak * 1e9) * 1e3, 5),
                "other_instructions_per_launch": {k[len("SQ_INSTS_"):].lower(): int(v) for k, v in e.items()
                                                  if k in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS",
                                                           "SQ_INSTS_BRANCH")}})
    return out


def algorithmic_bytes(cfg, n_shaded, n_ball_vertices):
    """B_alg = W*H*16 + N_shaded*4*M + sum_draws n_inst*(n_vert*44 + 128) + n_idx*4 + 6432 + 144."""
    m = 5 if cfg.enable_normal_map else 4
    frame = cfg.width * cfg.height * 16
    tex = n_shaded * 4 * m
    geom = cfg.n_instances * (n_ball_vertices * 44 + 128) + 1 * (4 * 44 + 128) + 6 * 4
    uniforms = 6432 + 144
    return {"total": frame + tex + geom + uniforms, "tile_kernel": frame + tex + uniforms, "geometry": geom}


class ClockSampler:
    """sclk / mclk / socket power of ONE GPU read from sysfs on a thread of its own (plain file reads: hwmon freq1_input,
    freq2_input, power1_input under /sys/class/drm/card*/device whose PCI address is the device's) -- no child process, no
    rocm-smi, nothing that touches the HIP runtime.  The thread sleeps between samples (the interpreter lock is free while it
    sleeps and while it reads), so it costs the frame loop a few tens of microseconds every `period_s`.  summary(t0, t1)
    gives min / mean / max over the samples taken inside a window of time.perf_counter()."""

    def __init__(self, pci_address, period_s=0.002):
        import glob
        import threading
        self.files, self.samples, self.period_s, self.read_us = {}, [], period_s, None
        self.card = None
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if os.path.basename(os.path.realpath(dev)).lower() != pci_address.lower():
                    continue
            except OSError:
                continue
            for key, name in (("sclk_mhz", "freq1_input"), ("mclk_mhz", "freq2_input"), ("power_w", "power1_input"),
                              ("power_w", "power1_average")):
                hits = glob.glob(os.path.join(dev, "hwmon", "hwmon*", name))
                if hits and key not in self.files and os.access(hits[0], os.R_OK):
                    self.files[key] = hits[0]
            self.card = dev
            break
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.files else None

    def _read(self):
        out = {}
        for key, f in self.files.items():
            try:
                v = float(open(f).read().split()[0])
            except (OSError, ValueError, IndexError):
                continue
            out[key] = v / 1e6   # Hz -> MHz, microwatt -> W
        return out

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            v = self._read()
            t1 = time.perf_counter()
            self.read_us = (t1 - t) * 1e6 if self.read_us is None else 0.9 * self.read_us + 0.1 * (t1 - t) * 1e6
            self.samples.append((0.5 * (t + t1), v))
            self._stop.wait(self.period_s)

    def start(self):
        if self._thread:
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread and self._thread.is_alive():
            self._thread.join(timeout=1.0)

    def summary(self, t0, t1):
        if not self.files:
            return {"available": False, "why": "no readable hwmon files for this device under /sys/class/drm"}
        inside = [v for (t, v) in self.samples if t0 <= t <= t1]
        if not inside:   # a region shorter than the sampling period: the nearest sample on either side
            before = [x for x in self.samples if x[0] < t0][-1:]
            after = [x for x in self.samples if x[0] > t1][:1]
            inside = [v for (_, v) in before + after]
        out = {"available": True, "samples": len(inside), "period_ms": self.period_s * 1e3,
               "read_us": round(self.read_us or 0.0, 1), "source": self.card}
        for key in ("sclk_mhz", "mclk_mhz", "power_w"):
            vals = [v[key] for v in inside if key in v]
            if vals:
                out[key] = {"min": round(min(vals), 1), "mean": round(sum(vals) / len(vals), 1), "max": round(max(vals), 1)}
        return out


def device_pci_address(index):
    """'0000:c1:00.0' of HIP device `index` (hipDeviceGetPCIBusId through the runtime torch has loaded); None if unavailable"""
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            return buf.value.decode()
    except OSError:
        pass
    return None


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup cpu.max / cfs quota), or None: a box may show 256 logical CPUs to
    sched_getaffinity and still be held to a share of them"""
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(period)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / period
    except (OSError, ValueError):
        return None


def cpu_baseline(cfg, maps, budget_s=20.0):
    """Oracle (CPU restatement, kind "port") on the host cores: bounded sample of the same workload.
    `value` = the oracle's threaded form (bbo_render_parallel: primitives set up once, in parallel, then 8-row bands from a
    queue -- SURVEY 8(d)(ii)) on every core this process may use: its affinity mask, held to the container's CPU quota when there
    is one (a one-GPU box shows 256 logical CPUs and grants 16).  Beside it the single-thread whole-frame run, whose frame the GPU
    is compared with, and -- for continuity with rounds 1-4 -- the band-per-call form on at most 16 threads."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import bbo, scenes

    mat = bbo.MaterialData(maps)
    sc = scenes.shaderball_scene(cfg, mat)
    W, H = sc.width, sc.height
    mpix = W * H / 1e6
    rgba = np.zeros((H, W, 4), np.float32)
    arr = (bbo.Draw * len(sc.draws))(*[d.c_struct() for d in sc.draws])
    L = bbo.lib()

    def band(b):
        st = bbo.Stats()
        rc = L.bbo_render(bbo._p(sc.frame), bbo._p(sc.view), arr, len(sc.draws), W, H, b[0], b[1], 0, bbo._p(rgba), None,
                          None, C.byref(st))
        assert rc == 0
        return st.n_shaded

    cores_available = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cpu_quota()
    cores = max(1, cores_available if quota is None else min(cores_available, int(np.ceil(quota))))
    # single thread, whole frame in one call (the scalar port as written)
    t0 = time.perf_counter()
    n1 = band((0, H))
    t_single = time.perf_counter() - t0
    # every usable core, the threaded oracle: at least five frames, until about half of the budget is used
    rgba_all = np.zeros((H, W, 4), np.float32)
    _, st_all = bbo.render_parallel(sc, threads=cores, rows=8, out=rgba_all)   # (threads started, pages touched)
    if st_all["n_shaded"] != n1 or not np.array_equal(rgba_all.view(np.uint32), rgba.view(np.uint32)):
        raise SystemExit("the threaded oracle renders another frame than the scalar one")
    reps, t_all = 0, 0.0
    while reps < 5 or (t_all * cores < 0.5 * budget_s and reps < 200):
        t0 = time.perf_counter()
        bbo.render_parallel(sc, threads=cores, rows=8, out=rgba_all)
        t_all += time.perf_counter() - t0
        reps += 1
    # rounds 1-4's form: one bbo_render call per 32-row band (each call sets every primitive up again) on <= 16 threads
    band_threads = max(1, min(cores_available, 16))
    bands = [(y, min(y + 32, H)) for y in range(0, H, 32)]
    b_reps, t_bands = 0, 0.0
    with ThreadPoolExecutor(band_threads) as ex:
        while b_reps < 3:
            t0 = time.perf_counter()
            n = sum(ex.map(band, bands))
            t_bands += time.perf_counter() - t0
            assert n == n1
            b_reps += 1
    return {"value": round(mpix * reps / t_all, 3), "unit": "Mpixels/s", "cores": cores, "cores_available": cores_available,
            "cpu_quota": quota, "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{reps} full {cfg.name} frame(s) ({W}x{H}), oracle/bb_oracle.c bbo_render_parallel, {cores} threads, 8-row bands "
                      f"from a queue ({t_all * cores:.1f} CPU-seconds); single-thread whole-frame run: {mpix / t_single:.3f} Mpixels/s",
            "single_thread_value": round(mpix / t_single, 3),
            "all_cores_value": round(mpix * reps / t_all, 3), "all_cores": cores, "all_cores_frames": reps,
            "all_cores_is": "= value: every core this process may use (affinity mask, held to the container's CPU quota `cpu_quota`)",
            "band_calls_value": round(mpix * b_reps / t_bands, 3), "band_calls_threads": band_threads,
            "band_calls_is": "rounds 1-4's figure: one bbo_render call per 32-row band, each repeating the set-up of every primitive",
            "n_shaded": int(n1)}, rgba


# ------------------------------------------------------------------------------------------------
# N > 1: every attempt runs in FRESH child processes under a supervisor that never touches torch or the GPU
# ------------------------------------------------------------------------------------------------
# The exchange of an N > 1 run is the one part of this file no single-GPU box can execute (RCCL between two devices, peer
# stores over xGMI).  If it hangs on real hardware, the run must still end with a line: the supervisor watches the workers'
# milestones, kills the whole process group of an attempt that stops making progress (or exits non-zero) and starts the next
# rung of the ladder in NEW processes -- a process that has initialised the GPU is never re-executed or reused:
#     as asked (default: native, the library's own ncclAllGather)  ->  torch.distributed all-gather  ->  the same, plain RGBA32F
# The line of the attempt that completes carries `exchange.attempts`: what was tried before it and why it was given up.
# Two launch forms, one supervisor:
#   * `python bench.py --gpus N` (no launcher): the parent starts `torch.distributed.run ... bench.py ... --worker` per attempt;
#   * under a launcher (WORLD_SIZE set, the driver's N > 1 form): every rank process is the supervisor of its own worker, the
#     ranks agree on "this attempt failed" through a flag file, and each attempt meets on its own rendezvous port.
MILESTONES = ("spawned", "imported", "group", "first_frame", "warm", "timed", "done")
EXIT_FATAL = 3   # a worker's exit code for a failure that is not the exchange's: the supervisor does not try the next rung


def attempt_ladder(exchange, gather, present):
    """[(exchange, gather), ...]: the attempt as asked, then the fallbacks (each only once)"""
    first = (exchange or "native", gather)
    ladder = [first]
    for rung in (("torch", gather), ("torch", "rgba32f")):
        if rung not in ladder and not (present and rung[1] != gather):
            ladder.append(rung)
    return ladder


def stall_limit_s(waiting_for, steps, warmup, cpu_budget):
    """seconds without a new milestone before an attempt is given up (BBR_BENCH_STALL_LIMIT: one number for every phase)"""
    if os.environ.get("BBR_BENCH_STALL_LIMIT"):
        return float(os.environ["BBR_BENCH_STALL_LIMIT"])
    return {"imported": 300.0,                                   # the first `import torch` on a fresh box takes minutes
            "group": 120.0, "first_frame": 180.0,                # rendezvous; textures, uploads, communicator, first exchange
            "warm": 120.0 + 0.05 * warmup, "timed": 60.0 + 0.05 * steps,
            "done": 240.0 + 3.0 * cpu_budget}.get(waiting_for, 120.0)


def _kill_group(p):
    import signal
    for sig, wait in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            p.wait(timeout=wait)
            return
        except Exception:   # noqa: BLE001  (subprocess.TimeoutExpired: escalate)
            continue


# The attempt that is running, for the supervisor's signal handlers: an attempt lives in a session of its own (so that its whole
# process group can be killed when it stalls), which also means a signal sent to the supervisor's group does not reach it.
# Whoever ends the supervisor -- the driver's timeout, a launcher tearing its ranks down, Ctrl-C -- must end the attempt too,
# or its workers stay on the GPUs, possibly inside a hung collective.
_RUNNING = {"p": None}


def _end_running_attempt():
    p = _RUNNING.get("p")
    if p is not None and p.poll() is None:
        _kill_group(p)
    _RUNNING["p"] = None


def _install_supervisor_handlers():
    import atexit
    import signal

    def on_signal(signum, _frame):
        print(f"[bench supervisor] signal {signum}: ending the running attempt's process group", file=sys.stderr, flush=True)
        _end_running_attempt()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sig, on_signal)
        except (ValueError, OSError):   # (not the main thread / not permitted: atexit still covers an orderly end)
            pass
    atexit.register(_end_running_attempt)


def _die_with_parent():
    """preexec of an attempt's first process: SIGTERM when the supervisor dies without running its handlers (SIGKILL, a crash)"""
    try:
        libc = C.CDLL("libc.so.6", use_errno=True)
        libc.prctl(1, 15, 0, 0, 0)   # PR_SET_PDEATHSIG, SIGTERM
    except OSError:
        pass


def run_attempt(cmd, env, watch_ranks, limits, fail_flag=None, deadline=None):
    """One attempt in its own session (process group).  Returns (ok, stdout_text, why, fatal): ok = every watched rank reached
    "done" (or the group exited 0); why = what was seen otherwise; fatal = a worker reported a failure that no other exchange
    can cure (its message), the ladder stops.  `deadline` (time.monotonic()): the attempt is given up when it passes."""
    import subprocess
    import tempfile
    import threading
    fd, ms_path = tempfile.mkstemp(prefix="bbr_bench_ms_")
    os.close(fd)
    env = dict(env, BBR_BENCH_MILESTONES=ms_path)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True, preexec_fn=_die_with_parent)
    _RUNNING["p"] = p
    chunks = []
    t = threading.Thread(target=lambda: chunks.append(p.stdout.read()), daemon=True)
    t.start()
    stage, last, why = 0, time.monotonic(), None
    reached = {}
    while True:
        rc = p.poll()
        try:
            for line in open(ms_path).read().splitlines():
                parts = line.split()
                if len(parts) == 2 and parts[1] in MILESTONES:
                    reached[parts[0]] = max(reached.get(parts[0], 0), MILESTONES.index(parts[1]))
        except OSError:
            pass
        now_stage = min([reached.get(str(r), 0) for r in watch_ranks]) if watch_ranks else 0
        if now_stage > stage:
            stage, last = now_stage, time.monotonic()
        if rc is not None:
            break
        if stage == len(MILESTONES) - 1:
            try:
                p.wait(timeout=60.0)     # everything is done: teardown only
            except Exception:            # noqa: BLE001
                _kill_group(p)
            break
        waiting_for = MILESTONES[stage + 1]
        if time.monotonic() - last > limits(waiting_for):
            why = f"no progress for {limits(waiting_for):.0f} s while waiting for milestone '{waiting_for}' (ranks at {dict(sorted(reached.items()))})"
            _kill_group(p)
            break
        if deadline is not None and time.monotonic() > deadline:
            why = f"the run's wall-time limit passed while waiting for milestone '{waiting_for}' (ranks at {dict(sorted(reached.items()))})"
            _kill_group(p)
            break
        if fail_flag and os.path.exists(fail_flag):
            why = "another rank gave this attempt up"
            _kill_group(p)
            break
        time.sleep(0.2)
    _RUNNING["p"] = None
    t.join(timeout=5.0)
    fatal = None
    try:
        fatal = open(ms_path + ".fatal").read().strip() or None
    except OSError:
        pass
    for f in (ms_path, ms_path + ".fatal"):
        try:
            os.unlink(f)
        except OSError:
            pass
    out = (chunks[0] if chunks else b"").decode(errors="replace")
    ok = why is None and fatal is None and (stage == len(MILESTONES) - 1 or p.returncode == 0)
    if p.returncode == EXIT_FATAL and fatal is None:
        fatal = "a worker left with the fatal exit code (its message is on stderr)"
    if not ok and why is None:
        why = fatal or f"exit code {p.returncode} after milestone '{MILESTONES[stage]}'"
    return ok, out, why, fatal


def _worker_argv(argv, exchange, gather):
    """the command line of an attempt: the caller's arguments with --exchange / --gather of this rung, marked --worker"""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("--exchange", "--gather"):
            skip = True
            continue
        if a.startswith("--exchange=") or a.startswith("--gather=") or a == "--worker":
            continue
        out.append(a)
    return out + ["--exchange", exchange, "--gather", gather, "--worker"]


def _emit_first_json_line(text):
    for line in text.splitlines():
        if line.startswith("{"):
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
            return True
    return False


def supervise(args, argv):
    """Run the N > 1 bench as a ladder of attempts (see above).  Returns the exit code.  Nothing here imports torch or
    touches a GPU."""
    import socket
    _install_supervisor_handlers()
    # (the one-GPU rehearsal gathers through gloo, where the library's RCCL communicator cannot come up with two ranks per device)
    first = args.exchange or ("native" if os.environ.get("BBR_BENCH_BACKEND", "nccl") == "nccl" else "torch")
    ladder = attempt_ladder(first, args.gather, args.present)
    limits = lambda name: stall_limit_s(name, args.steps, args.warmup, args.cpu_budget)   # noqa: E731
    under_launcher = "WORLD_SIZE" in os.environ
    rank = int(os.environ.get("RANK", "0"))
    base_env = dict(os.environ)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what this pool's driver supports (RCCL, peer pushes)
    base_env.setdefault("OMP_NUM_THREADS", "1")
    # the whole ladder has a wall-time limit below the driver's own (1500 s for an N > 1 run): whatever happens, the supervisor
    # ends by itself, having ended its workers
    deadline = time.monotonic() + float(os.environ.get("BBR_BENCH_WALL_LIMIT", "1300"))
    flag_base = None
    if under_launcher:
        port0 = int(os.environ.get("MASTER_PORT", "29533"))
        flag_base = os.path.join("/tmp", f"bbr_bench_{os.getppid()}_{port0}")
        if rank == 0:   # flags of an earlier run with the same parent and port must not decide this one
            for k in range(len(ladder)):
                for suffix in (f"_attempt{k}.failed", ".done"):
                    try:
                        os.unlink(flag_base + suffix)
                    except OSError:
                        pass
    done_flag = flag_base + ".done" if flag_base else None

    def cleanup_flags():
        # (the .failed flags only: the run's .done flag stays for the ranks that are still on their way -- a rank whose worker
        #  stalls in teardown looks for it a stall limit later -- and is removed by rank 0 of the next run with this parent and port)
        if flag_base and rank == 0:
            time.sleep(2.0)   # (the other ranks' supervisors look at them for a moment longer)
            for k in range(len(ladder)):
                try:
                    os.unlink(f"{flag_base}_attempt{k}.failed")
                except OSError:
                    pass
    attempts = []
    for k, (exchange, gather) in enumerate(ladder):
        env = dict(base_env, BBR_BENCH_ATTEMPTS=json.dumps(attempts))
        wargv = _worker_argv(argv, exchange, gather)
        flag = None
        if under_launcher:
            # this process is rank `rank` of a launcher's group: supervise OUR worker; attempt k meets on its own port
            env["MASTER_PORT"] = str(port0 + 1 + k)
            # (the launcher's agent hosts the store of ITS rendezvous and tells its children to use it; an attempt's workers
            #  meet on their own port, where rank 0's worker has to host the store itself)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            flag = f"{flag_base}_attempt{k}.failed"
            cmd = [sys.executable, os.path.abspath(__file__)] + wargv
            watch = [rank]
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + wargv
            watch = list(range(args.gpus))
        sys.stdout.flush()
        ok, out, why, fatal = run_attempt(cmd, env, watch, limits, flag, deadline)
        if ok:
            if rank == 0 and not _emit_first_json_line(out):
                ok, why = False, "the attempt ended without a JSON line"
            else:
                if done_flag:
                    try:
                        open(done_flag, "w").close()   # "this run has its line": see below
                    except OSError:
                        pass
                cleanup_flags()
                return 0
        # The ranks agree on SUCCESS too: if some rank's worker reached "done" -- rank 0's has then printed the
        # line -- a rank whose own worker merely stalled in teardown must not walk the remaining rungs alone against
        # rendezvous that can never complete.
        if done_flag and os.path.exists(done_flag):
            print(f"[bench supervisor rank {rank}] attempt {k + 1} ended here with '{why}', but another rank's worker completed it "
                  "(the line is printed): leaving with 0", file=sys.stderr, flush=True)
            return 0
        attempts.append({"exchange": exchange, "gather": gather, "gave_up_because": why})
        print(f"[bench supervisor rank {rank}] attempt {k + 1}/{len(ladder)} ({exchange}, {gather}) given up: {why}", file=sys.stderr, flush=True)
        if flag:
            try:
                open(flag, "w").close()   # tell the other ranks' supervisors
            except OSError:
                pass
            time.sleep(1.0)               # let them see it before the next attempt's rendezvous starts
        if fatal:
            print(f"[bench supervisor rank {rank}] not an exchange failure -- no further attempt: {fatal}", file=sys.stderr, flush=True)
            cleanup_flags()
            return 1
        if time.monotonic() > deadline:
            print(f"[bench supervisor rank {rank}] wall-time limit reached: no further attempt", file=sys.stderr, flush=True)
            break
    print(f"[bench supervisor rank {rank}] every attempt failed: {attempts}", file=sys.stderr, flush=True)
    cleanup_flags()
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5"])
    ap.add_argument("--tile-mode", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--band-rows", type=int, default=0, help="screen band height for N > 1 (0 = tile height)")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="HIP events around k_shade on every n-th step of the timed region; 0 = ten samples spread over it (events on every step cost "
                         "about 6 %% of the frame rate they are there to describe: each pair keeps consecutive k_shade launches "
                         "from overlapping head to tail)")
    ap.add_argument("--frames-in-flight", type=int, default=None, choices=[1, 2, 3, 4],
                    help="frames queued on the GPU at once (the reference keeps 2).  Default 3, and 4 for c2: a 1080p frame is a "
                         "chain of dependent kernels ~95 us long that fills a fraction of the GPU, so its rate is chain length / "
                         "frames in flight (C2, round 4: 45 / 31 / 24 us per frame with 2 / 3 / 4); at 4K three frames fill the "
                         "machine (C3, round 4: 120 / 108 / 108 us with 2 / 3 / 4 on the round's first build; "
                         "profiles/r04_scheduling_experiments.txt)")
    ap.add_argument("--stream-layout", type=int, default=2, choices=[0, 1, 2],
                    help="option stream_layout of the library (include/bibim_hip.h); 2 (one stream per frame slot) is its default")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="any other option of the library (bbr_set_option), e.g. --opt no_tail_items=0; recorded in config.options")
    ap.add_argument("--gather", default="packed", choices=["packed", "rgba32f", "rgba16f"],
                    help="N > 1: what the all-gather moves -- the shard as rgb + one alpha bit per pixel (lossless, 12.1 B per "
                         "pixel; default) or as plain RGBA32F (16 B); the reassembled frame is the same, bit for bit.  rgba16f: "
                         "every channel rounded to binary16, the reference's own HDR attachment format (8 B per pixel, LOSSY: a "
                         "different output, reported as such in config.output, never the default)")
    ap.add_argument("--no-clock-samples", action="store_true",
                    help="do not sample this GPU's sclk / mclk / socket power from sysfs during the run (`clocks` in the line)")
    ap.add_argument("--clock-period-ms", type=float, default=2.0, help="sampling period of the clock thread")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the 1080p figure the default C3 run adds to its line (`also.c2_1080p`): the counter passes of "
                         "tools/profile_round.sh average every launch of a kernel in the process")
    ap.add_argument("--worker", action="store_true",
                    help="(internal) this process is one rank of one attempt, started by the supervisor (see `supervise`)")
    ap.add_argument("--exchange", default=None, choices=["torch", "native", "peer"],
                    help="N > 1: who runs the exchange.  native (default): bbr_allgather_frame -- the library packs, calls "
                         "ncclAllGather (RCCL over xGMI) on the frame's own stream and un-interleaves (the id travels through "
                         "torch.distributed once).  torch: torch.distributed all_gather_into_tensor (RCCL) on a stream of the "
                         "harness + the library's un-interleave kernel.  peer: bbr_push_shard -- one kernel stores this rank's "
                         "block into every rank's gather buffer (IPC handles; all xGMI links at once), a host barrier orders "
                         "the landing.  None of the three has run on more than one GPU (DESIGN.md section 5): every N > 1 run is "
                         "a ladder of attempts in fresh processes (as asked -> torch -> torch rgba32f), see `supervise`")
    ap.add_argument("--force-dist", action="store_true", help="exercise the all-gather path with WORLD_SIZE = 1")
    ap.add_argument("--render-pass", default="forward", choices=["forward", "deferred"],
                    help="forward_brdf.* (the path BASELINE measures) or the reference's deferred path, gbuffer.* + brdf.*")
    ap.add_argument("--present-fused", action="store_true",
                    help="with --present: the kernels write presented RGBA8 pixels themselves (option present_fused), no fp32 "
                         "frame and no separate presentation pass")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region, compare the last gathered frame with the same frame rendered unpartitioned "
                         "on this rank's GPU (bit for bit)")
    ap.add_argument("--present", action="store_true",
                    help="every step also runs the presentation step (tone map + sRGB + RGBA8, SURVEY 8(f) rank 1); "
                         "for N > 1 the RGBA8 shards are gathered instead of the fp32 ones (a quarter of the payload)")
    args = ap.parse_args()
    if args.frames_in_flight is None:
        args.frames_in_flight = 4 if args.workload == "c2" else 3

    # N > 1: this process is a SUPERVISOR unless it was started as a worker (--worker).  It starts the ranks of one attempt
    # after the other as CHILD processes -- before it has imported torch or made any HIP call (a process that has touched
    # the GPU must never be replaced or re-executed on this pool) -- and relays the JSON line of the attempt that completes.
    if args.gpus > 1 and not args.worker:
        raise SystemExit(supervise(args, sys.argv[1:]))

    # milestones for the supervisor (N > 1) and the rehearsal hook that makes a rank stall at one of them
    ms_path = os.environ.get("BBR_BENCH_MILESTONES")
    my_rank = int(os.environ.get("RANK", "0"))

    def mark(name):
        hook = os.environ.get("BBR_BENCH_STALL", "").split(":")   # "exchange:rank:milestone", e.g. native:1:first_frame
        if len(hook) == 3 and hook[0] == (args.exchange or "native") and hook[1] == str(my_rank) and hook[2] == name:
            print(f"[bench rank {my_rank}] BBR_BENCH_STALL: stalling in front of '{name}'", file=sys.stderr, flush=True)
            time.sleep(1e6)
        if ms_path:
            with open(ms_path, "a") as f:
                f.write(f"{my_rank} {name}\n")
    mark("spawned")

    def fatal(message):
        """a failure no other rung of the ladder can cure (parity, shaded-pixel count ...): say so where the supervisor looks, and
        leave with EXIT_FATAL on every rank"""
        print(f"[bench rank {my_rank}] FATAL: {message}", file=sys.stderr, flush=True)
        if ms_path:
            try:
                with open(ms_path + ".fatal", "a") as f:
                    f.write(f"rank {my_rank}: {message}\n")
            except OSError:
                pass
        sys.stderr.flush()
        os._exit(EXIT_FATAL)   # (not SystemExit: a process group that is being torn down must not wait in destructors of collectives)

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when its
    # communicator comes up): from here on file descriptor 1 is stderr, and the JSON line goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    mark("imported")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    # Rehearsal hooks (not used by the driver): BBR_BENCH_BACKEND=gloo gathers through host memory, and
    # BBR_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0, so that the N > 1 code path (partition, shard buffers,
    # events, un-interleave) can be run as real separate processes on a one-GPU box, where RCCL refuses two ranks per GPU.
    backend = os.environ.get("BBR_BENCH_BACKEND", "nccl")
    if args.exchange is None:   # (the gloo rehearsal on one GPU cannot open an RCCL communicator with two ranks per device)
        args.exchange = "native" if backend == "nccl" else "torch"
    if os.environ.get("BBR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    n_cus = int(torch.cuda.get_device_properties(local_rank).multi_processor_count)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    mark("group")

    from bibim_renderer_amd import Renderer, configs, textures
    from bibim_renderer_amd import scene as S

    cfg = configs.CONFIGS[args.workload]
    maps = textures.make_material(cfg.texture_size)
    ball = S.load_shaderball_vertices()

    r = Renderer(cfg.width, cfg.height, device=local_rank)
    if args.tile_mode is not None:
        r.set_option("tile_mode", args.tile_mode)
    r.set_option("frames_in_flight", args.frames_in_flight)
    r.set_option("stream_layout", args.stream_layout)
    for o in args.opt:
        r.set_option(o.split("=")[0], int(o.split("=")[1]))
    r.set_option("render_pass", 1 if args.render_pass == "deferred" else 0)
    if args.present_fused:
        if not args.present:
            raise SystemExit("--present-fused needs --present")
        r.set_option("present_fused", 1)
    material = r.upload_material(maps)
    scene, cam, settings = S.config_scene(r, cfg, ball)

    W, H = cfg.width, cfg.height
    dist_path = world > 1 or args.force_dist
    if dist_path:
        # Each rank renders its interleaved screen bands into a compact shard; RCCL all-gathers the shards over xGMI
        # and a copy kernel un-interleaves them into the row-major frame.  Every buffer exists twice, so that the exchange
        # of frame N (pack, all-gather, un-interleave, on a stream of their own) overlaps the rendering of frame N+1; the
        # shard is handed back to the renderer as soon as the pack kernel (or the gather) has read it.
        band_rows = args.band_rows or r.tile_height()
        r.set_partition(rank, world, band_rows)
        shard_rows = r.shard_rows()
        shard_t = [torch.empty((shard_rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered_t = [torch.empty((world * shard_rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]  # [rank][shard row]
        frame_t = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        packed = args.gather == "packed" and not args.present
        half = args.gather == "rgba16f" and not args.present
        from bibim_renderer_amd import partition as P
        form = P.SHARD_RGBA8 if args.present else (P.SHARD_PACKED if packed else (P.SHARD_RGBA16F if half else P.SHARD_RGBA32F))
        exchange_note = None
        if args.exchange == "native":
            # the 128-byte id: made by rank 0's library, handed round by the harness, one ncclCommInitRank per rank.  If any
            # rank cannot open its communicator (no librccl for dlopen, ...) ALL ranks fall back to the torch exchange -- a
            # collective must be entered by everybody or by nobody.
            # First a vote on something that is NOT collective -- can this rank load librccl and make an id at all? -- and only
            # if every rank says yes do they enter ncclCommInitRank together: a rank that failed locally and went straight to
            # the vote would leave the others waiting inside the collective.
            ok, why, box = 1, "", [None]
            try:
                r.comm_probe()
                if rank == 0:
                    box = [r.comm_unique_id()]
            except Exception as e:   # noqa: BLE001  (reported below)
                ok, why = 0, f"bbr_comm_probe / bbr_comm_unique_id: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                if world > 1:
                    dist.broadcast_object_list(box, src=0)
                try:
                    r.comm_init(rank, world, box[0])     # collective; a failure here is the supervisor's to catch (stall / exit code)
                except Exception as e:   # noqa: BLE001
                    ok, why = 0, f"bbr_comm_init: {e}"
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                if world > 1:
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                exchange_note = f"native exchange unavailable on some rank ({why or 'another rank'}): torch.distributed all-gather instead"
                print(f"[bench rank {rank}] {exchange_note}", file=sys.stderr, flush=True)
                args.exchange = "torch"
        if args.exchange == "peer":
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            block = r.exchange_block_bytes(form)
            mine = []
            for _ in range(2):   # plain hipMalloc allocations: their IPC handles open in the peers as they are
                ptr = ctypes.c_void_p()
                if hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(block * world)) != 0:
                    raise SystemExit("hipMalloc of a gather buffer failed")
                mine.append(ptr.value)
            handles = [None] * world
            dist.all_gather_object(handles, (local_rank, [r.ipc_export(p) for p in mine]))
            peer_devs = [h[0] for h in handles]
            peer_ptrs = [[mine[b] if p == rank else r.ipc_open(handles[p][1][b]) for p in range(world)] for b in range(2)]
            pushed = [torch.cuda.Event(), torch.cuda.Event()]
            pending = []
        if packed or half:   # (a staged block: made by the library from the shard, un-made by bbr_unpack_whole)
            pb = r.exchange_block_bytes(form)
            packed_t = [torch.empty((pb,), dtype=torch.uint8, device="cuda") for _ in range(2)]
            gathered_packed_t = [torch.empty((world * pb,), dtype=torch.uint8, device="cuda") for _ in range(2)]
        if args.present:
            shard8_t = [torch.empty((shard_rows, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
            gathered8_t = [torch.empty((world * shard_rows, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
            frame8_t = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        # (one stream for pack + gather + un-interleave: with the un-interleave on a stream of its own the link would
        #  never wait for it, but on one GPU the extra stream cost 40 % of the frame rate -- see DESIGN.md on stream
        #  counts -- and what it does to eight ranks could not be measured here)
        ag_stream = torch.cuda.Stream()
        up_stream = ag_stream
        consumed = [torch.cuda.Event(), torch.cuda.Event()]    # the shard has been read (by the pack kernel / the gather)
        gathered_ev = [torch.cuda.Event(), torch.cuda.Event()]  # the gathered buffer is complete
        unpacked = [torch.cuda.Event(), torch.cuda.Event()]     # ... and has been un-interleaved: free for gather n+2
    step_no = [0]
    submit_frame = S.frame_call(r, scene, cam, settings, material)   # (draw_frame with its arguments marshalled once)

    def step():
        n = step_no[0]
        step_no[0] = n + 1
        if not dist_path:
            submit_frame()  # asynchronous; internal double-buffered framebuffer
            if args.present:
                r.present()
            return
        b = n & 1
        if args.exchange == "native":
            # library-owned shard, gather buffer and whole frame per frame slot; the exchange is the frame's last step on
            # the slot's own stream, so the next frame of the slot is ordered behind it and the other slots render meanwhile
            S.draw_frame(r, scene, cam, settings, material)
            if args.present:
                r.present()
            r.allgather_frame(form)
            return
        if args.exchange == "peer":
            S.draw_frame(r, scene, cam, settings, material)
            if args.present:
                r.present()
            if pending:
                finish_peer(pending.pop())     # frame n - 1: everyone's blocks have landed -> un-interleave; the GPU renders n meanwhile
            r.push_shard(form, peer_ptrs[b], peer_devs)
            r.stream_wait_frame(ag_stream.cuda_stream)
            pushed[b].record(ag_stream)
            pending.append(n)
            return
        if n >= 2:
            r.wait_event(consumed[b].cuda_event)             # shard[b] is free again once gather n-2 has read it
        r.set_output_device_ptr(shard_t[b].data_ptr(), shard_t[b].numel() * 4)
        S.draw_frame(r, scene, cam, settings, material)
        if args.present:
            r.present(shard8_t[b].data_ptr())
        r.stream_wait_frame(ag_stream.cuda_stream)
        with torch.cuda.stream(ag_stream):
            if packed or half:
                r.stage_shard(form, packed_t[b].data_ptr(), ag_stream.cuda_stream)
                consumed[b].record(ag_stream)
            src, dst = (shard8_t[b], gathered8_t[b]) if args.present else (
                (packed_t[b], gathered_packed_t[b]) if (packed or half) else (shard_t[b], gathered_t[b]))
            if n >= 2:
                ag_stream.wait_event(unpacked[b])                # un-interleave n-2 has finished reading dst
            if backend == "nccl":
                dist.all_gather_into_tensor(dst, src)
            else:  # rehearsal: through host memory
                h_src = src.cpu()
                h_dst = torch.empty(dst.shape, dtype=dst.dtype)
                dist.all_gather_into_tensor(h_dst, h_src)
                dst.copy_(h_dst)
            if not (packed or half):
                consumed[b].record(ag_stream)
            gathered_ev[b].record(ag_stream)
        with torch.cuda.stream(up_stream):
            up_stream.wait_event(gathered_ev[b])
            if args.present:
                r.unpack_gathered_rgba8(gathered8_t[b].data_ptr(), frame8_t[b].data_ptr(), up_stream.cuda_stream)
            elif packed or half:
                r.unpack_whole(form, gathered_packed_t[b].data_ptr(), frame_t[b].data_ptr(), up_stream.cuda_stream)
            else:
                r.unpack_gathered(gathered_t[b].data_ptr(), frame_t[b].data_ptr(), up_stream.cuda_stream)
            unpacked[b].record(up_stream)

    def finish_peer(m):
        b = m & 1
        pushed[b].synchronize()            # this rank's copies of frame m have landed in every peer
        if m >= 1:
            unpacked[b ^ 1].synchronize()  # ... and its un-interleave of frame m - 1 is done: after the barrier the peers
        dist.barrier()                     #     overwrite that buffer with frame m + 1.  Barrier: so have everyone's
        whole = frame8_t[b] if args.present else frame_t[b]
        r.unpack_whole(form, peer_ptrs[b][rank], whole.data_ptr(), ag_stream.cuda_stream)
        unpacked[b].record(ag_stream)

    def fence():
        if dist_path and args.exchange == "peer" and pending:
            finish_peer(pending.pop())
        r.synchronize()
        torch.cuda.synchronize()
        if dist is not None:   # (--force-dist runs the same collectives with one rank)
            dist.barrier()
            torch.cuda.synchronize()

    # first frame sizes every capacity (bins etc.); synchronize() re-renders if one overflowed
    step()
    fence()
    mark("first_frame")
    stats = r.stats()
    # The host side of a step is one ctypes call into the C++ shim; a cyclic-GC pass of the interpreter (tens of ms with
    # torch's object graph loaded) inside the timed region would be a stall of the harness, not of the renderer.  Collect
    # now -- the frames below bring the GPU back up to speed afterwards -- and keep the collector off until the end.
    import gc
    gc.collect()
    gc.disable()
    # The clocks of this GPU, sampled from sysfs on a thread of its own from here to the end of the measurements
    clocks = None
    if not args.no_clock_samples:
        addr = device_pci_address(local_rank)
        clocks = ClockSampler(addr, args.clock_period_ms * 1e-3).start() if addr else None
    # a few dozen frames let every frame slot learn its item count (the shading launch of a slot is sized from the slot's
    # previous frame) ...
    for _ in range(40):
        step()
    fence()
    layout, layout_decided, layout_ms = r.stream_layout_state()
    use_events = not args.no_timing_events
    # HIP events around k_shade on sampled steps of the timed region: ten samples of a long run, but never closer than every
    # fourth step (a pair on every step costs ~6 % of the rate it describes; the driver's 20-step run gets five samples)
    event_stride = max(1, min(args.event_stride, args.steps)) if args.event_stride > 0 else max(min(4, max(1, args.steps // 2)), args.steps // 10)
    # Everything that synchronises or idles the GPU happens BEFORE the quarter second of frames below: switching the timing
    # events on (the library drains and creates 2560 events: the GPU idles for a millisecond, and its clocks need ~100 ms
    # of load to come back -- with the switch BEHIND those frames the default run read 109-110 us per C3 frame, 102 with
    # --warmup 200, 100.7 without events: gpurun_out/r4/cmp1.txt, cmp3.txt), the statistics read-back.  Between the last
    # warm-up step and the clock there is only what the contract asks for -- barrier + synchronize -- so the timed steps start
    # on a GPU that was busy a moment ago, as the frames of a running application do.
    fence()
    if use_events:
        # timed region: only the two events that bracket the dominant kernel (the full five-event breakdown costs
        # 2-3 % of the frame rate; it is taken in the one-frame-in-flight pass after the timed region)
        r.set_option("timing", 2)
        r.set_option("timing_stride", event_stride)
    fence()
    # ... and then the GPU is WARMED UNTIL ITS RATE IS STEADY (round 5): blocks of 50 frames, each closed by a device
    # synchronisation, until two consecutive blocks agree within 1 % -- at most two seconds.  Round 4 rendered for a fixed quarter
    # of a second; the driver's fresh-lease run read 116 us per frame where this harness's boxes read 103-110, and nothing in the
    # line could say whether the GPU had still been coming up.  The block history is in the line now (`warm_up`).  Untimed, like
    # the W warm-up steps that follow; the timed region is still exactly K steps.  (N > 1: every step holds collectives, so the
    # ranks decide together -- the verdict is all-reduced -- and render the same number of blocks.)
    warm_blocks, warm_why, t_warm0, warm_spans = [], None, time.perf_counter(), []
    WARM_BLOCK, WARM_TOL, WARM_LIMIT_S, WARM_MIN_BLOCKS = 50, 0.01, 2.0, 3
    while True:
        tb = time.perf_counter()
        for _ in range(WARM_BLOCK):
            step()
        if dist_path and args.exchange == "peer" and pending:
            finish_peer(pending.pop())
        torch.cuda.synchronize()
        warm_blocks.append((time.perf_counter() - tb) / WARM_BLOCK * 1e3)
        warm_spans.append((tb, time.perf_counter()))
        steady = (len(warm_blocks) >= WARM_MIN_BLOCKS and
                  abs(warm_blocks[-1] - warm_blocks[-2]) <= WARM_TOL * warm_blocks[-2])
        out_of_time = time.perf_counter() - t_warm0 > WARM_LIMIT_S or len(warm_blocks) >= 400
        stop = steady or out_of_time
        if dist is not None:
            v = torch.tensor([1 if stop else 0, 1 if steady else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(v, op=dist.ReduceOp.MIN)   # everybody steady (or out of time) -> stop together
            stop, steady = bool(int(v[0].item())), bool(int(v[1].item()))
        if stop:
            warm_why = "two consecutive blocks within 1 %" if steady else f"time limit ({WARM_LIMIT_S} s)"
            break
    for _ in range(args.warmup):
        step()
    if dist_path and args.exchange == "peer" and pending:
        finish_peer(pending.pop())
    # One guard for a whole class of hangs (round 4 found one by accident: ranks that had computed different numbers of
    # pre-warm frames around collectives): every trip count around a collective must be the same on every rank.  Checked
    # HERE, in front of the timed region, with one MIN and one MAX reduction -- a mismatch ends the run with a message instead of
    # a stall that the supervisor can only time out.
    if dist is not None:
        torch.cuda.synchronize()   # (the library's own collectives of the warm-up steps are done before torch's communicator is used)
        trip = torch.tensor([args.steps, args.warmup, len(warm_blocks), event_stride, args.frames_in_flight, step_no[0]],
                            dtype=torch.int64, device="cuda")
        lo, hi = trip.clone(), trip.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"rank {rank}: the ranks disagree on a trip count around collectives -- (steps, warmup, warm-up blocks, "
                             f"event stride, frames in flight, frames so far) = {trip.tolist()} here, min {lo.tolist()}, max {hi.tolist()}")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    if use_events:
        r.timing_reset()   # (host-side after the synchronize: the samples of the warm-up steps are dropped)
    mark("warm")
    overflow_before = r.capacity_growths()   # host-side counter
    frames_before = step_no[0] + overflow_before   # frames submitted so far (incl. the re-renders after overflows)
    r.host_timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_submitted = time.perf_counter()
    # the closing bracket of the timed region: every stream of the device drained (hipDeviceSynchronize) and all ranks
    # there.  The library's own synchronising call additionally copies the frame's counter block to the host to look for
    # a capacity overflow; that check is not part of the K steps and runs right after the clock stops.
    if dist_path and args.exchange == "peer" and pending:
        finish_peer(pending.pop())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t_end = time.perf_counter()
    elapsed = t_end - t0
    host_t = r.host_timing()
    fence()
    mark("timed")
    if r.capacity_growths() != overflow_before:   # (overflows healed during the warm-up frames do not count)
        raise SystemExit("a capacity overflowed inside the timed region: the frames timed were incomplete")
    gc.enable()
    # The host's side of the timed region: if the host were the bottleneck it would never be blocked
    # on a frame slot; if the GPU is, the host spends what is left of every frame period blocked.
    host = {"submit_us_per_step": round(host_t["submit_ns"] / max(1, host_t["frames"]) / 1e3, 2),
            "blocked_us_per_step": round(host_t["blocked_ns"] / max(1, host_t["frames"]) / 1e3, 2),
            "blocked_steps": int(host_t["blocked_frames"]), "steps": int(host_t["frames"]),
            "loop_us_per_step": round((t_submitted - t0) / args.steps * 1e6, 2),
            "drain_us": round((t_end - t_submitted) * 1e6, 1),
            "is": "submit = inside bbr_end_frame (staging, launches), of which blocked = waiting for the frame slot's previous frame "
                  "(counted by the library, steady_clock, no GPU call); loop = the harness's step loop per step (Python + shim + "
                  "submit); drain = from the last submit to the end of the timed region"}
    clocks_timed = clocks.summary(t0, t_end) if clocks else None

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ns = torch.tensor([stats["n_shaded"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(ns)
        n_shaded_total = int(ns.item())
    else:
        n_shaded_total = stats["n_shaded"]

    ms_per_step = elapsed / args.steps * 1e3
    value = W * H * args.steps / elapsed / 1e6

    roofline = None
    if use_events:
        n_ev, avg_frame_ms, avg_geom_ms, avg_raster_ms, avg_shade_ms = r.timing_summary()
        # Dominant kernel: k_shade.  Its algorithmic bytes (SURVEY.md 8(d) per-pixel figures x the pixels one launch
        # shades on this rank): 16 B framebuffer write + 4 B x M texel reads per shaded pixel, + the uniform blocks.
        # The background pixels' 16 B are written by k_raster, vertex streams are read by k_geometry; the whole frame's
        # B_alg against the whole step time is reported beside it (frame_*).
        m = 5 if cfg.enable_normal_map else 4
        shade_bytes = stats["n_shaded"] * (16 + 4 * m) + 6432 + 144
        achieved = shade_bytes / (avg_shade_ms * 1e-3) / 1e9 if avg_shade_ms > 0 else 0.0
        balg = algorithmic_bytes(cfg, n_shaded_total, ball.shape[0])
        # (the committed PMC passes are of the unpartitioned launch: no figure for a rank's share of the frame)
        traffic, traffic_src = profiled_traffic(args.workload) if world == 1 and not args.force_dist else (None, None)
        roofline = {"bound": "hbm", "kernel": "k_shade", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(shade_bytes), "avg_kernel_ms": round(avg_shade_ms, 5),
                    "launches_timed": int(n_ev), "event_stride": event_stride,
                    "frames_in_flight": args.frames_in_flight, "frames_before_timed_region": int(frames_before),
                    "frame_algorithmic_bytes": int(balg["total"]),
                    "frame_achieved_gbs": round(balg["total"] / (ms_per_step * 1e-3) / 1e9, 2),
                    "frame_frac": round(balg["total"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "n_shaded": int(n_shaded_total),
                    "limiter": None}   # filled in below from this run's own figures
        lines_bytes, lines_src = committed_texel_lines(args.workload)
        roofline["traffic_is"] = ("bytes through the L2s' memory side (FETCH_SIZE x 2 + WRITE_SIZE): Infinity-Cache hits are included -- gfx950 "
                                  "exposes no counter behind the L2 -- so it is an UPPER bound of the DRAM bytes")
        if lines_bytes:
            roofline["distinct_texel_line_bytes"] = lines_bytes
            roofline["distinct_texel_line_source"] = lines_src
        t_ms, t_alone_ms, t_src = committed_trace_ms(args.workload)
        roofline["avg_kernel_ms_is"] = (f"HIP events on the kernel's own stream right in front of and behind k_shade's main launch, with "
                                        f"{args.frames_in_flight} frames in flight: the kernel shares the GPU with the other frames' kernels, so this is "
                                        "its duration UNDER THAT LOAD (trace_kernel_ms is the same quantity from the kernel trace; "
                                        "one_frame_in_flight / trace_kernel_alone_ms the kernel alone)")
        if not t_ms and t_src:
            roofline["trace_source"] = t_src   # (why there are no trace_* figures: the committed trace is of other kernels)
        if t_ms:
            roofline.update({"trace_kernel_ms": round(t_ms, 5), "trace_kernel_alone_ms": round(t_alone_ms, 5), "trace_source": t_src,
                             "trace_frac": round(shade_bytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "trace_alone_frac": round(shade_bytes / (t_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        if world == 1 and not args.force_dist:
            roofline["valu"] = valu_roofline(args.workload, "k_shade", avg_shade_ms, n_cus)
        if dist is not None:
            # every rank's own figures (its bands of the frame), gathered to rank 0's line: k_shade inside the timed region,
            # and -- after it, one frame in flight, no exchange -- what each of the rank's three kernels costs by itself
            r.set_option("frames_in_flight", 1)
            r.set_option("timing_stride", 1)
            r.set_option("timing", 1)
            if not (args.exchange == "native"):
                r.set_output_device_ptr(shard_t[0].data_ptr(), shard_t[0].numel() * 4)
            for _ in range(4):
                S.draw_frame(r, scene, cam, settings, material)
            r.synchronize()
            r.timing_reset()
            for _ in range(20):
                S.draw_frame(r, scene, cam, settings, material)
            r.synchronize()
            _, f1r, g1r, ra1r, s1r = r.timing_summary()
            r.set_option("timing", 0)
            r.set_option("frames_in_flight", args.frames_in_flight)
            mine = {"rank": rank, "device": local_rank, "n_shaded": int(stats["n_shaded"]), "avg_kernel_ms": round(avg_shade_ms, 5),
                    "algorithmic_bytes_per_launch": int(shade_bytes), "achieved": round(achieved, 2),
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "alone": {"avg_geometry_ms": round(g1r, 5), "avg_raster_ms": round(ra1r, 5), "avg_shade_ms": round(s1r, 5),
                              "avg_device_frame_latency_ms": round(f1r, 5)}}
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            roofline["per_rank"] = per_rank

    if roofline is not None and not dist_path and not args.present_fused:
        # Outside the timed region: the same kernels with one frame in flight, i.e. without the other frame's
        # geometry/raster sharing the CUs -- what each kernel costs by itself (the timed region above overlaps them).
        r.set_option("frames_in_flight", 1)
        r.set_option("timing_stride", 1)
        r.set_option("timing", 1)
        for _ in range(5):
            step()
        fence()
        r.timing_reset()
        for _ in range(min(args.steps, 50)):
            step()
        fence()
        n1, f1, g1, ra1, s1 = r.timing_summary()
        roofline["one_frame_in_flight"] = {
            "launches_timed": int(n1), "avg_kernel_ms": round(s1, 5), "avg_geometry_ms": round(g1, 5),
            "avg_raster_ms": round(ra1, 5), "avg_device_frame_latency_ms": round(f1, 5),
            "achieved": round(shade_bytes / (s1 * 1e-3) / 1e9, 2) if s1 > 0 else 0.0,
            "frac": round(shade_bytes / (s1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if s1 > 0 else 0.0}
        v = roofline.get("valu")
        if v and v.get("valu_instructions_per_launch") and s1 > 0:
            a1 = v["valu_instructions_per_launch"] / (s1 * 1e-3) / 1e9
            v["one_frame_in_flight"] = {"avg_kernel_ms": round(s1, 5), "achieved": round(a1, 1), "frac": round(a1 / v["peak"], 4),
                                        "fp32_tflops": round(v["fp32_flop_per_launch"] / (s1 * 1e-3) / 1e12, 2),
                                        "fp32_frac": round(v["fp32_flop_per_launch"] / (s1 * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)}
        # what limits the kernel, from THIS run: its vector-ALU issue floor (committed instruction counts of this workload and
        # these kernel sources / 2-cycle issue) against its duration alone on the GPU; no figure when no such summary exists
        if v and v.get("issue_floor_ms") and s1 > 0:
            floor = v["issue_floor_ms"]
            alone = roofline.get("trace_kernel_alone_ms") or s1     # (the kernel trace's duration when a summary of these kernels exists)
            sclk = ((clocks_timed or {}).get("sclk_mhz") or {}).get("mean")
            text = (f"k_shade alone {alone * 1e3:.1f} us ({'kernel trace' if roofline.get('trace_kernel_alone_ms') else 'HIP events'}) against a "
                    f"vector-ALU issue floor of {floor * 1e3:.1f} us at 2 cycles per wave64 instruction and the nominal "
                    f"{SHADER_CLOCK_GHZ} GHz ({floor / alone:.2f})")
            if sclk:
                floor_sclk = floor * SHADER_CLOCK_GHZ * 1e3 / sclk
                v["issue_floor_ms_at_reported_sclk"] = round(floor_sclk, 5)
                text += (f"; {floor_sclk * 1e3:.1f} us at the {sclk:.0f} MHz this GPU reported over the timed region ({floor_sclk / alone:.2f}), "
                         "more with each instruction class at its measured issue cost (C3: + 14 %, profiles/*_k_shade_issue_floor.txt)")
            roofline["limiter"] = text + ("; vector issue, the vector-memory return path and the wave slots all stand near their limits "
                                          "(DESIGN.md section 3, k_shade); HBM is not the limiter")
        else:
            roofline["limiter"] = "no instruction-count summary of this workload on these kernel sources (profiles/*_pmc_sq.json)"
        # the reference's own setting, two frames in flight (src/main.cpp:38), and one frame's device latency
        r.set_option("timing", 0)
        r.set_option("frames_in_flight", 2)
        for _ in range(6):
            step()
        fence()
        n2 = max(10, min(args.steps, 100))
        r.host_timing_reset()
        t2a = time.perf_counter()
        for _ in range(n2):
            step()
        torch.cuda.synchronize()
        t2b = time.perf_counter()
        t2 = t2b - t2a
        h2 = r.host_timing()
        fence()
        roofline["frames_in_flight_2"] = {"ms_per_step": round(t2 / n2 * 1e3, 5), "steps": n2,
                                          "value": round(W * H * n2 / t2 / 1e6, 2), "unit": "Mpixels/s",
                                          "host": {"submit_us_per_step": round(h2["submit_ns"] / max(1, h2["frames"]) / 1e3, 2),
                                                   "blocked_us_per_step": round(h2["blocked_ns"] / max(1, h2["frames"]) / 1e3, 2),
                                                   "blocked_steps": int(h2["blocked_frames"])},
                                          "clocks": clocks.summary(t2a, t2b) if clocks else None,
                                          "what": "the same step with the reference's two frames in flight (numFrames, src/main.cpp:38)"}
        roofline["single_frame_device_latency_ms"] = round(f1, 5)
        r.set_option("timing", 1)
        # the next row of SURVEY 8(f), measured beside the path: k_present alone, 16 B read + 4 B written per pixel
        import dataclasses
        settings_tm = dataclasses.replace(settings, enable_tone_mapping=1, exposure=1.0)
        S.draw_frame(r, scene, cam, settings_tm, material)
        fence()
        r.timing_reset()
        for _ in range(50):
            r.present()
        fence()
        n_p, ms_p = r.present_timing()
        p_bytes = W * H * 20
        roofline["next_row_present"] = {
            "kernel": "k_present", "bound": "hbm", "launches_timed": int(n_p), "avg_kernel_ms": round(ms_p, 5),
            "algorithmic_bytes_per_launch": p_bytes, "achieved": round(p_bytes / (ms_p * 1e-3) / 1e9, 2) if ms_p > 0 else 0.0,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(p_bytes / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_p > 0 else 0.0,
            "what": "binary16 HDR -> tone map (on, exposure 1) -> sRGB UNORM8 of the whole frame"}
        r.set_option("frames_in_flight", args.frames_in_flight)

    # north_star asks for 1080p AND 4K figures: after the timed region of the default (C3) run, C2 -- ShaderBall, one point
    # light, 1920x1080 -- for a hundred steps with ITS four frames in flight, in a context of its own
    also = None
    if args.workload == "c3" and not dist_path and not args.present and args.render_pass == "forward" and not args.opt and not args.no_also:
        cfg2 = configs.CONFIGS["c2"]
        r.synchronize()
        rb = Renderer(cfg2.width, cfg2.height, device=local_rank)
        rb.set_option("frames_in_flight", 4)
        mb = rb.upload_material(textures.make_material(cfg2.texture_size))
        sb, cb, setb = S.config_scene(rb, cfg2, ball)
        S.draw_frame(rb, sb, cb, setb, mb)
        rb.synchronize()
        submit_c2 = S.frame_call(rb, sb, cb, setb, mb)
        for _ in range(60):
            submit_c2()
        rb.synchronize()
        torch.cuda.synchronize()
        n_c2 = 100
        tb = time.perf_counter()
        for _ in range(n_c2):
            submit_c2()
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        rb.synchronize()
        also = {"c2_1080p": {"workload": f"{cfg2.name}: {cfg2.description}", "steps": n_c2, "frames_in_flight": 4,
                             "ms_per_step": round(tb / n_c2 * 1e3, 5), "value": round(cfg2.width * cfg2.height * n_c2 / tb / 1e6, 2),
                             "unit": "Mpixels/s", "n_shaded": int(rb.stats()["n_shaded"])}}
        sb.close()
        rb.close()

    if clocks:
        clocks.stop()
    verified = None
    if args.verify and dist_path:
        last = (step_no[0] - 1) & 1
        torch.cuda.synchronize()
        if args.exchange == "native":
            got = r.read_whole_frame(form)
        else:
            got = (frame8_t[last] if args.present else frame_t[last]).cpu().numpy()
        r2 = Renderer(cfg.width, cfg.height, device=local_rank)
        r2.set_option("render_pass", 1 if args.render_pass == "deferred" else 0)
        m2 = r2.upload_material(maps)
        scene2, cam2, settings2 = S.config_scene(r2, cfg, ball)
        S.draw_frame(r2, scene2, cam2, settings2, m2)
        if args.present:
            r2.present()
            want = r2.read_presented()
        else:
            want = r2.read_framebuffer()
            if half:   # the binary16 wire form: every channel of the unpartitioned frame to the nearest binary16 value
                with np.errstate(over="ignore"):
                    want = want.astype(np.float16).astype(np.float32)
        verified = bool(np.array_equal(got.view(np.uint8), want.view(np.uint8)))
        if not verified and os.environ.get("BBR_BENCH_DEBUG"):
            S.draw_frame(r2, scene2, cam2, settings2, m2)
            if args.present:
                r2.present()
                want2 = r2.read_presented()
            else:
                want2 = r2.read_framebuffer()
            bad = (got.view(np.uint8) != want.view(np.uint8)).reshape(H, W, -1).any(axis=2)
            ys, xs = np.nonzero(bad)
            print(f"[debug rank {rank}] want == want2: {np.array_equal(want2.view(np.uint8), want.view(np.uint8))}; "
                  f"got == want2: {np.array_equal(got.view(np.uint8), want2.view(np.uint8))}; differing pixels {bad.sum()} "
                  f"x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}; sample got {got[ys[0], xs[0]]} want {want[ys[0], xs[0]]} "
                  f"want2 {want2[ys[0], xs[0]]}", flush=True)
        scene2.close(); r2.close()
        if not verified:
            bad = (got.view(np.uint8) != want.view(np.uint8)).reshape(H, -1).any(axis=1)
            rows = np.nonzero(bad)[0]
            extra = ""
            if not args.present:
                d = np.abs(got.astype(np.float64) - want.astype(np.float64))
                extra = f"; max |diff| {np.nanmax(d):.3g}, differing values {int((got.view(np.uint32) != want.view(np.uint32)).sum())}"
            raise SystemExit(f"rank {rank}: gathered frame differs from the unpartitioned render in {rows.size} rows "
                             f"(first {rows[:8].tolist()}, last {rows[-3:].tolist()}){extra}")

    cpu, parity, parity_literal = None, None, None
    # Rank 0 times the CPU baseline and checks the frame against the oracle while the other ranks wait.  If that fails, it
    # fails for EVERY rank, at once and with the message (the others sat in the final barrier until the
    # supervisor timed them out, and the ladder then repeated a failure that no other exchange could cure).
    failure = None
    try:
        if rank == 0 and not args.no_cpu_baseline:   # (N > 1: on rank 0's host cores, while the other ranks wait at the final barrier)
            cpu, oracle_frame = cpu_baseline(cfg, maps, args.cpu_budget)
            if cpu["n_shaded"] != n_shaded_total:
                raise SystemExit(f"GPU shaded {n_shaded_total} pixels, oracle {cpu['n_shaded']}: parity broken")
            # the frame the bench has been rendering, against the frame the oracle just rendered (forward pass only: the
            # CPU baseline is the forward oracle)
            if args.render_pass == "forward" and not dist_path and not args.present_fused:
                gpu_frame = r.read_framebuffer()
                tol = 1e-4 * np.maximum(1.0, np.abs(oracle_frame))
                d = np.abs(gpu_frame - oracle_frame)
                parity = {"bit_exact": bool(np.array_equal(gpu_frame.view(np.uint32), oracle_frame.view(np.uint32))),
                          "within_1e-4_times_max_1_ref": bool((d <= tol).all()), "max_abs_diff": float(np.nanmax(d)),
                          "pixels": int(W * H)}
                if not parity["within_1e-4_times_max_1_ref"]:
                    raise SystemExit(f"GPU frame outside the 1e-4 tolerance of the oracle: {parity}")
                parity["oracle_form"] = ("contract: the kernel's own evaluation order of the light loop (oracle/bb_oracle.c "
                                         "light_surface_contract); bit_exact here is NOT parity with the GLSL -- see parity_vs_literal")
                # ... and against the LITERAL form: forward_brdf.frag:29-70 / brdf.glsl statement by statement
                from oracle import bbo, scenes
                lit, _ = bbo.render_bands(scenes.shaderball_scene(cfg, bbo.MaterialData(maps)), flags=bbo.FLAG_LITERAL)
                dl = np.abs(gpu_frame.astype(np.float64) - lit.astype(np.float64))
                parity_literal = {"oracle_form": "literal: the GLSL statement by statement (BBO_FLAG_LITERAL)",
                                  "tolerance": "absolute 1e-4 per channel (BASELINE.json); also 1e-4 * max(1, |ref|) (BASELINE.md)",
                                  "within_abs_1e-4": bool((dl <= 1e-4).all()),
                                  "within_1e-4_times_max_1_ref": bool((dl <= 1e-4 * np.maximum(1.0, np.abs(lit))).all()),
                                  "max_abs_diff": float(np.nanmax(dl)), "max_abs_ref": float(np.nanmax(np.abs(lit))),
                                  "bit_exact": bool(np.array_equal(gpu_frame.view(np.uint32), lit.view(np.uint32))), "pixels": int(W * H)}
                if not parity_literal["within_1e-4_times_max_1_ref"]:
                    raise SystemExit(f"GPU frame outside the 1e-4 tolerance of the literal GLSL form: {parity_literal}")
    except SystemExit as e:   # (the checks above end with a message)
        failure = str(e.code)
    except Exception as e:    # noqa: BLE001
        failure = f"{type(e).__name__}: {e}"
    if dist is not None:
        box = [failure]
        dist.broadcast_object_list(box, src=0)
        failure = box[0]
    if failure:
        fatal(failure)

    if rank == 0:
        out = {
            "metric": "Mpixels/sec shaded (4K PBR ShaderBall)" if args.workload == "c3" else f"Mpixels/sec shaded ({cfg.name})",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {cfg.description}", "width": W, "height": H,
                       "instances": cfg.n_instances, "lights": len(cfg.lights), "triangles": int(stats["n_prims"]),
                       "textures": f"{cfg.texture_size}x{cfg.texture_size} RGBA8 x5 (seed 0x5EED)",
                       "partition": "single GPU" if world == 1 else f"interleaved {args.band_rows or r.tile_height()}-row bands, "
                                    f"{world} ranks, " + {"torch": "ncclAllGather (torch.distributed) of ",
                                                          "native": "bbr_allgather_frame (ncclAllGather on the frame's stream) of ",
                                                          "peer": "bbr_push_shard (one kernel storing to every rank's buffer; copies if a peer cannot be mapped) of "}[args.exchange] + (
                                        "RGBA8 shards" if args.present else
                                        "shards packed as rgb + alpha bit (lossless, 12.1 B/pixel)" if args.gather == "packed"
                                        else "shards rounded to binary16 (8 B/pixel, lossy)" if args.gather == "rgba16f"
                                        else "RGBA32F shards") + " + un-interleave",
                       "tile": f"{stats['tile_w']}x{stats['tile_h']}",
                       "output": ("presented RGBA8 (fused)" if args.present_fused else "RGBA32F frame + presented RGBA8")
                                 if args.present else ("RGBA32F frame" if not (dist_path and half) else
                                                       "frame of binary16 values (R16G16B16A16_SFLOAT, the reference's HDR attachment format, "
                                                       "src/render.h:94), widened to RGBA32F -- LOSSY, not the default metric's output"),
                       "render_pass": args.render_pass,
                       "stream_layout": layout, **({"options": args.opt} if args.opt else {})},
            "roofline": roofline, "cpu_baseline": cpu,
            "host": host, "clocks": clocks_timed,
            "warm_up": {"block_steps": WARM_BLOCK, "blocks_ms_per_step": [round(x, 5) for x in warm_blocks],
                        "stopped_because": warm_why, "frames": WARM_BLOCK * len(warm_blocks),
                        "clocks_first_block": clocks.summary(*warm_spans[0]) if clocks else None,
                        "clocks_last_block": clocks.summary(*warm_spans[-1]) if clocks else None,
                        "rule": f"blocks of {WARM_BLOCK} steps until two consecutive blocks agree within {WARM_TOL:.0%} (at least "
                                f"{WARM_MIN_BLOCKS} blocks, at most {WARM_LIMIT_S} s), then the {args.warmup} warm-up steps"},
            # `value` is W*H / t: every pixel of the frame is produced, but only N_shaded of them run the PBR
            # shader (the rest get the clear colour from k_raster); the shaded-only rate:
            "shaded_mpixels_per_s": round(n_shaded_total * args.steps / elapsed / 1e6, 2),
            "shaded_fraction_of_frame": round(n_shaded_total / float(W * H), 4),
        }
        if dist_path:
            block_bytes = int(r.exchange_block_bytes(form))
            # what the links allow, so that a reader of a scaling record can hold the measurement against it: every rank has to
            # RECEIVE world - 1 blocks per frame.  On the node's full mesh each of them can arrive over its own xGMI link (the
            # direct pattern: bbr_push_shard, or RCCL when it picks a direct algorithm): block / link rate; a ring passes every
            # block through every link in turn: (world - 1) x block / link rate.
            link = XGMI_LINK_GBS_PER_DIRECTION
            direct_ms = block_bytes / (link * 1e9) * 1e3 if world > 1 else 0.0
            ring_ms = direct_ms * (world - 1)
            rccl_ranks = None
            if args.exchange == "native":
                try:
                    rccl_ranks = r.comm_count()   # ncclCommCount on the library's own communicator
                except Exception as e:            # noqa: BLE001
                    rccl_ranks = f"unavailable: {e}"
            try:
                attempts_before = json.loads(os.environ.get("BBR_BENCH_ATTEMPTS", "[]"))
            except ValueError:
                attempts_before = []
            out["exchange"] = {"who": args.exchange, "form": {P.SHARD_RGBA8: "rgba8", P.SHARD_PACKED: "packed rgb + alpha bit",
                                                              P.SHARD_RGBA32F: "rgba32f", P.SHARD_RGBA16F: "rgba16f"}[form],
                               "bytes_per_rank_block": block_bytes,
                               "bytes_received_per_rank_per_frame": block_bytes * (world - 1),
                               "ranks_in_communicator": int(dist.get_world_size()),
                               "ranks_in_communicator_is": "torch.distributed's world size; rccl_ranks is what the library's own communicator reports",
                               "rccl_ranks": rccl_ranks, "backend": backend,
                               "band_rows": int(args.band_rows or r.tile_height()), "shard_rows": int(r.shard_rows()),
                               "link_rate_gbs_assumed": link,
                               "link_rate_is": "xGMI, one direction of one link: half of the 153.6 GB/s a link carries both ways; seven links per GPU",
                               "link_bound_ms_estimate": round(direct_ms, 4), "link_bound_ms_estimate_ring": round(ring_ms, 4),
                               "link_bound_mpixels_per_s_estimate": (round(W * H / (direct_ms * 1e-3) / 1e6, 1) if direct_ms > 0 else None),
                               "link_bound_is": "block / link rate: every rank receives world - 1 blocks per frame, each over its own link of the "
                                                "full mesh at best (ring: x (world - 1)); with frames in flight the exchange of frame n overlaps "
                                                "the rendering of n + 1, so the frame period is at least max(this, a rank's rendering time)",
                               "attempts": attempts_before,
                               **({"note": exchange_note} if exchange_note else {})}
        if also is not None:
            out["also"] = also
        if verified is not None:
            out["verified_against_unpartitioned_render"] = verified
        if parity is not None:
            out["parity_vs_oracle"] = parity
        if parity_literal is not None:
            out["parity_vs_literal"] = parity_literal
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.barrier()   # (rank 0 may have spent a while on the CPU baseline)
    mark("done")
    scene.close()
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
