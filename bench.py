#!/usr/bin/env python
"""Headline benchmark: denoising-step throughput of Next-DiT 2B (Lumina-Next, BASELINE.json configs[1]) at
1024x1024 with classifier-free guidance, flow-matching Euler ODE, bf16, synthetic data / random-init weights.

    python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 the script launches its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (the driver's form; RANK / WORLD_SIZE come from the launcher)

A "step" is one denoising step = one ``forward_with_cfg`` (NFE) on a cond+uncond pair plus the Euler update,
driven exactly as a user would: ``Sampler(transport).sample_ode(...)(z, model.forward_with_cfg, **kwargs)``.
Each rank denoises ONE image (weak scaling: images shard over GPUs, no data-path collective; the text
features are broadcast once from rank 0 over RCCL before the timed region).  Inputs are resident in HBM when
the timed region starts.  Rank 0 prints one JSON line; ``value`` is whole-job latent-tokens/s
(images x latent tokens x NFE / wall), ``denoising_steps_per_s`` the same in NFE/s.

``--workload cfg4`` runs BASELINE configs[3] instead (Lumina-Next-SFT 2B GQA at 2048x2048 = 16384 latent tokens per image,
time-aware RoPE scaling, one image per GPU: its BASELINE form "batch = 8 sharded over 8 GPUs" is ``--gpus 8``).

``roofline`` is measured live with HIP events around every launch of the dominant kernel class (the bf16 MFMA GEMMs,
76 % of the algorithmic FLOPs) on the launch stream; ``cpu_baseline`` times the CPU oracle (fp32, host
cores) on ONE complete forward_with_cfg of the bench workload (all 24 layers: 30-70 s of CPU time, measured, not
extrapolated); the TRUE reference timing (unmodified reference module, authoring container, min of 3 with the host load
stated) is quoted beside it from profiles/.  ``power`` is the socket power sampled over the timed region.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md:42
TEXT_LEN = 128

WORKLOADS = {
    # name: (constructor, gqa, resolution, extra model kwargs, description)
    "cfg2": dict(ctor="NextDiT_2B_patch2", gqa=False, res=1024, scale_factor=1.0, scale_watershed=1.0,
                 desc="BASELINE configs[1]: Lumina-Next-T2I 2B (NextDiT_2B_patch2, d2304 L24 H32 hd72 F6144), 1024x1024 (4096 latent tokens)",
                 parity="tests/golden/full_2b.npz (one NFE, 24 layers) and full_2b_traj_euler30 / _midpoint10 (whole trajectories): outputs of the "
                        "UNMODIFIED reference module + sampler, gate 1.5 x the reference's own bf16-vs-fp32 distance (tests/test_gpu_fulldepth.py)"),
    "cfg2-gqa": dict(ctor="NextDiT_2B_GQA_patch2", gqa=True, res=1024, scale_factor=1.0, scale_watershed=1.0,
                     desc="Lumina-Next 2B GQA (NextDiT_2B_GQA_patch2, 32 / 8 heads) at the configs[1] workload, 1024x1024 (4096 latent tokens)",
                     parity="tests/golden/full_2b_gqa_ntk.npz: outputs of the unmodified reference module (mini package, GQA) at 4096 tokens"),
    "cfg4": dict(ctor="NextDiT_2B_GQA_patch2", gqa=True, res=2048, scale_factor=2.0, scale_watershed=0.3,
                 desc="BASELINE configs[3]: Lumina-Next-SFT 2B (NextDiT_2B_GQA_patch2), 2048x2048 any-resolution (16384 latent tokens, "
                      "NTK-aware / time-aware RoPE scale_factor 2, watershed 0.3)",
                 parity="tests/golden/full_2b_gqa_16k.npz: refbf16_* = outputs of the UNMODIFIED reference module in bf16 at 16 384 tokens (its flash "
                        "branch, round 6); the fp32 side is the restatement (the fp32 reference needs a 69 GB N x N mask), which equals the reference "
                        "bit for bit on the same weights at 4096 tokens (full_2b_gqa_ntk)"),
}


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(argv, n_gpus, port=None):
    """the torch.distributed.run command line `python bench.py --gpus N ...` re-executes itself under (one rank per GPU)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args, argv) -> None:
    """`python bench.py --gpus N` (N > 1) without a launcher environment: start N ranks and relay rank 0's JSON line."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    if args.share_device:
        env["LUMINA_SHARE_DEVICE"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = subprocess.run(launcher_command(argv, args.gpus), env=env).returncode
    sys.exit(rc)


def random_init_(model, seed):
    """Random-init weights of the named architecture directly on the GPU.  The reference zero-inits the adaLN /
    final / cap-embedder / gate paths (model.py:567,643,652,709,201), which would turn every block into an
    identity; give them the SURVEY.md 8d synthetic statistics instead so the timed work is representative."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("attention.gate"):
                p.normal_(0.0, 0.5, generator=g)
            elif p.dim() == 1 and name.endswith(".weight"):
                p.copy_(1.0 + 0.02 * torch.randn(p.shape, device=p.device, generator=g))
            elif p.dim() == 1:
                p.normal_(0.0, 0.02, generator=g)
            else:
                p.normal_(0.0, min(0.06, p.shape[-1] ** -0.5), generator=g)


class PowerSampler:
    """Socket power of one GPU sampled on a host thread while a region runs (amdsmi, else the hwmon power file).  The GEMM / attention
    kernels of this path run under the board's power cap; this is the measurement in watts beside the clock
    evidence: `avg_w`, `max_w`, the gfx clock seen, and joules per algorithmic TFLOP of the region."""

    def __init__(self, index=0, period_s=0.005):
        self.index, self.period, self.samples, self.clocks = index, period_s, [], []
        self._stop = self._thread = self._read = self._clk = None
        self.source = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[index]

            def read():
                d = amdsmi.amdsmi_get_power_info(h)
                for k in ("current_socket_power", "average_socket_power", "socket_power"):
                    v = d.get(k)
                    if isinstance(v, (int, float)) and v > 0:
                        return float(v)
                return None

            def clk():
                try:
                    c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                    v = c.get("clk", c.get("cur_clk"))
                    return float(v) if isinstance(v, (int, float)) else None
                except Exception:
                    return None

            if read() is not None:
                self._read, self._clk, self.source = read, clk, "amdsmi socket power"
        except Exception:
            pass
        if self._read is None:
            import glob
            files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") +
                           glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
            if files:
                f = files[min(index, len(files) - 1)]

                def read():
                    try:
                        with open(f) as fh:
                            return float(fh.read().strip()) * 1e-6
                    except Exception:
                        return None

                if read():
                    self._read, self._clk, self.source = read, (lambda: None), f

    def __enter__(self):
        import threading
        if self._read is None:
            return self
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                w = self._read()
                if w:
                    self.samples.append(w)
                c = self._clk()
                if c:
                    self.clocks.append(c)
                self._stop.wait(self.period)

        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()

    def report(self, seconds, tflop):
        if not self.samples:
            return None
        avg = sum(self.samples) / len(self.samples)
        out = {"avg_w": avg, "max_w": max(self.samples), "samples": len(self.samples), "source": self.source,
               "joule_per_tflop": avg * seconds / tflop if tflop > 0 else None}
        if self.clocks:
            out["gfx_clock_mhz_avg"] = sum(self.clocks) / len(self.clocks)
        return out


def pmc_summary(name):
    """A committed rocprofv3 --pmc summary of this same command (scripts/gpu_prof.sh -> profiles/rNN/<name>; PMC passes are separate runs by
    construction).  Returns (summary dict or None, provenance dict): the summary carries the sha256 of the kernel sources it was taken on
    (scripts/summarize_pmc.py) and the commit (scripts/stamp_profile.py); it is REFUSED - traffic null, reason stated - when one of those
    sources has changed since, so a stale counter can never sit beside a live timing."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", name)))
    if not files:
        return None, {"file": None, "status": "no committed summary"}
    with open(files[-1]) as f:
        d = json.load(f)
    prov = {"file": os.path.relpath(files[-1], REPO), "git_head": d.get("git_head"), "measured_in_this_run": False,
            "box": "builder's profiling box (a committed file), NOT this run's box"}
    stamps = d.get("source_sha256")
    if not stamps:
        prov["status"] = "refused: the summary carries no source stamp (taken before round 6)"
        return None, prov
    changed = []
    for fn, want in stamps.items():
        try:
            with open(os.path.join(REPO, "lumina-t2x_amd", "csrc", fn), "rb") as f:
                if hashlib.sha256(f.read()).hexdigest() != want:
                    changed.append(fn)
        except OSError:
            changed.append(fn)
    if changed:
        prov["status"] = "refused: " + ", ".join(changed) + " changed since the counters were collected"
        return None, prov
    prov["status"] = "sources unchanged since the counters were collected"
    return d, prov


def rocprof_gemm_time_per_nfe():
    """cross-check of the live HIP-event figure: GEMM-class kernel time per NFE from the committed rocprofv3 --kernel-trace --stats
    summary of this same command (profiles/rNN/rocprofv3_kernel_stats_rNN.csv), or None"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "rocprofv3_kernel_stats_r*.csv")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        rows = list(csv.DictReader(f))
    nfe = [int(r["Calls"]) for r in rows if "unpatchify" in r["Name"]]
    if not nfe or not nfe[0]:
        return None, None
    ns = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_bf16" in r["Name"])
    return ns / nfe[0] * 1e-6, os.path.relpath(files[-1], REPO)


def reference_cpu_timing():
    """the one TRUE reference number: 1 NFE of the unmodified NextDiT_2B_patch2 (fp32, CPU) in the authoring container, recorded
    by oracle/make_fulldepth_golden.py (the GPU box has no /root/reference)"""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "reference_cpu_timing.json")))
    if not files:
        return None
    path = files[-1]
    with open(path) as f:
        return json.load(f)


def physical_cores():
    """physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo); os.cpu_count() counts SMT siblings, which only
    slow an fp32 GEMM / SDPA mix down"""
    try:
        pairs, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    pairs.add((phys, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


class _cpu_only_cuda_calls:
    """the reference hard-codes `.cuda()` on its RoPE table (model.py:952) and moves it back with `.to(x.device)` (model.py:773); for the
    CPU timing the call is a no-op (as in oracle/ref_harness.py), restored on exit so that nothing else in the process is affected"""

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t

    def __exit__(self, *exc):
        import torch
        torch.Tensor.cuda = self._orig


def cpu_baseline(latent, n_tokens, _cfg=None):
    """The CPU leg of the bench line (north_star: "the reference timed on the box's host CPU cores in the same run").

    kind "reference": the UNMODIFIED reference module - lumina_next_t2i/models/model.py NextDiT_2B_patch2, imported from /root/reference in
    the authoring container or from the byte-identical files of the travelling archive oracle/_ref/reference_files.tar (oracle/build_ref.py;
    sha256-verified on extraction) behind
    the stub packages - fp32, host cores, ONE complete forward_with_cfg of the bench workload (24 of 24 layers, 4096 tokens, T = 128, CFG
    pair), timed, not extrapolated.  kind "port": the same call through the restatement (oracle/nextdit_oracle.py) when no copy travelled.
    Threads: one reference LAYER (the real SDPA + GEMM mix, same shapes) is timed at 8 / 16 / 32 / 64 threads capped at the physical core
    count, the fastest setting runs the full call; the host's 1-minute load average is recorded beside it."""
    import importlib
    import torch
    from oracle import ref_harness as R
    from oracle import synth

    root = R.timing_root()
    kw = dict(base_seqlen=n_tokens, proportional_attn=True)
    cfg = _cfg or synth.NextDiTConfig()  # (_cfg: the CPU test suite runs this function on a tiny model)
    assert _cfg is not None or (cfg.n_layers == 24 and cfg.dim == 2304)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(latent, latent), text_len=TEXT_LEN, uncond_len=8, seed=1)
    ncpu, nphys = os.cpu_count() or 1, physical_cores()

    def load():
        try:
            return os.getloadavg()[0]
        except OSError:
            return None

    if root:
        R.load_reference("lumina_next_t2i", root=root)
        M = importlib.import_module("models.model")

        def build(c, sd):
            m = M.NextDiT(**c.ctor_kwargs()).eval()
            r = m.load_state_dict(sd, strict=True, assign=True)
            assert not r.missing_keys and not r.unexpected_keys
            return lambda: m.forward_with_cfg(z, t, cap, mask, 4.0, **kw)
    else:
        from oracle import nextdit_oracle as O

        def build(c, sd):
            return lambda: O.forward_with_cfg(sd, c, z, t, cap, mask, cfg_scale=4.0, **kw)

    with torch.no_grad(), _cpu_only_cuda_calls():
        # thread count: one layer of the same width at the same token count, 1 warm-up + 1 timed call per setting
        cfg1 = synth.NextDiTConfig(**dict(cfg.to_dict(), n_layers=1))
        call1 = build(cfg1, synth.synth_state_dict(cfg1, seed=0, streams=True))
        probe, cores, best = {}, 1, float("inf")
        for n in sorted({min(nphys, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(n)
            call1()
            t0 = time.time()
            call1()
            probe[n] = time.time() - t0
            if probe[n] < best * 0.95:
                best, cores = probe[n], n
        del call1
        torch.set_num_threads(cores)
        t0 = time.time()
        sd = synth.synth_state_dict(cfg, seed=0, streams=True)
        t_draw = time.time() - t0
        call = build(cfg, sd)
        l0 = load()
        t0 = time.time()
        out = call()
        t_nfe = time.time() - t0
    assert torch.isfinite(out).all()
    kind = "reference" if root else "port"
    what = ("UNMODIFIED reference lumina_next_t2i/models/model.py NextDiT.forward_with_cfg (" + ("checkout " + root if R.available() else
            "byte-identical files from the archive oracle/_ref/reference_files.tar, every sha256 verified") + ")") if root else "oracle restatement nextdit_oracle.forward_with_cfg (no reference copy on this box)"
    res = {
        "value": n_tokens / t_nfe, "unit": "latent-tokens/s", "cores": cores, "kind": kind,
        "sample": (f"ONE complete fp32 forward_with_cfg of the bench workload ({cfg.n_layers} of {cfg.n_layers} layers, d={cfg.dim}, N={n_tokens}, T=128, B=2) through the {what}: "
                   f"{t_nfe:.1f} s measured on {cores} threads ({nphys} physical cores, {ncpu} logical CPUs, 1-min load before the call {l0}); "
                   f"weight draw {t_draw:.0f} s not counted"),
        "denoising_steps_per_s": 1.0 / t_nfe, "seconds_per_nfe": t_nfe, "load_1min": l0,
        "thread_probe_one_layer_s": {str(k): round(v, 3) for k, v in probe.items()},
    }
    ref = reference_cpu_timing()
    if ref:
        res["reference_module_authoring_container"] = ref
    return res


def box_probe(lib, torch, power):
    """Which box did this run draw?  Boxes of this pool differ by up to 10 % under identical code (29.0-30.2 vs 32.9-34.5 ms per step in
    round 4's own runs), more than a round usually moves.  Two readings, both independent of the engine's kernels' scheduling:
      * `avg_w` - socket power averaged over the timed region.  This is what separates the classes: every slow box of round 4 drew
        1173-1203 W under the step, every fast one ~1285 W (calibration table in profiles/r04/NOTES_same_box_numbers.md) - an
        MFMA-bound step is power-bound, so a box that sustains fewer watts is slower in proportion.  It is a
        continuum more than two classes (a 30.4 ms box drew 1208 W), and it is not the configured cap: `power_cap_w` (what amdsmi
        reports) read 1400 W on that box.  `class` cuts at 1240 W.
      * `probe_tflops` - a fixed GEMM (8192 x 12288 x 2304, random bf16, plain epilogue) run 40 times after the timed regions.  It reads
        1198-1220 on BOTH classes: 15 ms of one kernel do not reach the limit a 30-ms mixed step lives under.  Kept as the evidence
        that the classes do not differ in what a single kernel can do."""
    import ctypes as C
    if os.environ.get("LT_NO_EVENT_PROFILE"):  # a rocprofv3 run (scripts/gpu_prof.sh): keep the probe's launches out of the kernel statistics
        avg_w = (power or {}).get("avg_w")
        return {"avg_w": avg_w, "power_cap_w": None, "class": None if avg_w is None else ("fast" if avg_w >= 1240.0 else "slow"), "class_cut_w": 1240.0,
                "probe": "skipped under rocprofv3", "probe_tflops": None}
    M, N, K = 8192, 12288, 2304
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())

    def launch(n):
        for _ in range(n):
            rc = lib.lt_op_gemm_bf16(P(a), P(w), None, 1, P(c), M, N, K, 0, 0, st)
            assert rc == 0
    launch(10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(40)
    e1.record()
    torch.cuda.synchronize()
    tf = 2.0 * M * N * K * 40 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    avg_w = (power or {}).get("avg_w")
    cap = None
    try:
        import amdsmi
        h = amdsmi.amdsmi_get_processor_handles()[0]
        info = amdsmi.amdsmi_get_power_cap_info(h)
        cap = info.get("power_cap")
        if isinstance(cap, (int, float)) and cap > 10000:
            cap = cap / 1e6  # microwatts on some amdsmi versions
    except Exception:
        pass
    return {"avg_w": avg_w, "power_cap_w": cap, "class": None if avg_w is None else ("fast" if avg_w >= 1240.0 else "slow"), "class_cut_w": 1240.0,
            "probe": "gemm 8192x12288x2304 bf16, 40 launches back to back", "probe_tflops": tf}


def algorithmic_gemm_bytes_per_launch(M, d, F, dkv):
    """A + W + C bytes (bf16) of the four GEMM launches of a layer, averaged: the compulsory traffic the PMC figure is read against"""
    qkv = M * d + (d + 2 * dkv) * d + M * (d + 2 * dkv)
    o = M * d + d * d + M * d
    w13 = M * d + 2 * F * d + M * F  # the SwiGLU epilogue writes F columns
    w2 = M * F + d * F + M * d
    return 2.0 * (qkv + o + w13 + w2) / 4.0


def gemm_kernel_label(lib, M, d, F, dkv, tokens, hd):
    """names of the kernels the engine's dispatcher runs for this workload's GEMM shapes (lt_op_gemm_describe), and the number of
    GEMM launches per layer (4 when the QKV projection is one launch, else 5)"""
    import ctypes as C
    fused = bool(lib.lt_op_gemm_qkv_fusable(M, d + 2 * dkv, d, d + dkv, tokens, hd))
    shapes = [("O", M, d, d, 0), ("W1|W3+SwiGLU", M, 2 * F, d, 1), ("W2", M, d, F, 0)]
    if not fused:
        shapes = [("QK", M, d + dkv, d, 0), ("V", M, dkv, d, 2)] + shapes
    parts = ["QKV: gemm_bf16_w4q<3,9> (persistent 4 waves, 16x16x32 MFMA, 256x288, fused QKV: plain Q|K tiles + V^T tiles)"] if fused else []
    for nm, m, n, k, epi in shapes:
        buf = C.create_string_buffer(160)
        lib.lt_op_gemm_describe(m, n, k, epi, 0, buf, 160)
        parts.append(f"{nm}: {buf.value.decode()}")
    return "; ".join(parts), (4 if fused else 5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=29, help="timed denoising steps (NFE); 29 = one 30-point Euler grid")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="engine A/B knob name=value (lt_set_option), repeatable")
    ap.add_argument("--event-steps", type=float, default=0.5,
                    help="NFEs of the timed region whose GEMM launches carry HIP start/stop events (0 = all).  A timed launch "
                         "costs ~0.1 ms of queue idle time on this stack (the host waits on the dispatch signal), so the default "
                         "samples the first half NFE = embedder + every GEMM shape x 12 layers (every layer and every NFE has the "
                         "same shapes)")
    ap.add_argument("--profile-classes", type=int, default=1,
                    help="bit mask of kernel classes bracketed by HIP events IN THE TIMED REGION: 1 GEMM (the roofline kernel, "
                         "default), 2 attention, 4 other.  Every bracketed launch costs two event packets (all classes: +4 %% "
                         "wall), so the attention / other breakdown is taken in a short untimed pass after the timed region")
    ap.add_argument("--share-device", action="store_true",
                    help="with --gpus N > 1: all N ranks drive GPU 0 and the collectives run over gloo with a host hop.  Exercises the "
                         "self-launch / rank / shard / max-over-ranks / JSON relay logic with the real engine on a ONE-GPU box; it is NOT "
                         "a scaling measurement (the line says n_gpus 1, ranks N, shared_device true) and says nothing about RCCL")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions run back to back in this process: `value` / `ms_per_step` are the FIRST (the contract's K steps); "
                         "`ms_per_step_repeats` lists all of them so that a round-over-round delta can be told from run-to-run noise")
    ap.add_argument("--gqa", action="store_true", help="shorthand for --workload cfg2-gqa")
    ap.add_argument("--attn-variant", type=int, default=None, help="A/B knob: 1 baseline, 2 VALU-diet, 3 ping-pong, 4 one wave per SIMD (default)")
    ap.add_argument("--gemm-variant", type=int, default=None, help="A/B knob: 0 auto (default), 1 256x256, 2 256x288")
    args = ap.parse_args()
    maybe_self_launch(args, sys.argv[1:])

    import torch

    import lumina_t2x_amd  # noqa: F401
    from lumina_t2x_amd import _lib, models, parallel
    from lumina_t2x_amd.flops import flops_per_nfe
    from lumina_t2x_amd.transport import Sampler, create_transport

    if args.gqa and args.workload == "cfg2":
        args.workload = "cfg2-gqa"
    wl = WORKLOADS[args.workload]
    res = wl["res"]
    latent = res // 8
    n_tokens = (latent // 2) ** 2
    if args.attn_variant is not None:
        _lib.check(_lib.load().lt_set_option(b"attention_variant", args.attn_variant))
    for opt in args.opt:
        k, v = opt.split("=")
        _lib.check(_lib.load().lt_set_option(k.encode(), int(v)))
    if args.gemm_variant is not None:
        _lib.check(_lib.load().lt_set_option(b"gemm_variant", args.gemm_variant))

    share = args.share_device or os.environ.get("LUMINA_SHARE_DEVICE", "0") == "1"
    rank, world, local = parallel.init_distributed("nccl" if args.gpus > 1 else None, share_device=share)
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)

    with torch.device(dev):
        model = getattr(models, wl["ctor"])(qk_norm=True, cap_feat_dim=2048).to(torch.bfloat16)
    random_init_(model, seed=0)
    model.eval()
    # every rank has drawn its 4 GB of weights before the first collective: start-up skew between ranks (random init, first import of
    # torch on a cold box) ends HERE, two barriers and the warm-up's weight upload away from the timed region
    torch.cuda.synchronize()
    parallel.barrier()

    # text features: rank 0 "runs the text encoder" (synthetic), one RCCL broadcast, each rank keeps its image
    n_img = world
    feats = mask = None
    if rank == 0:
        g = torch.Generator(device="cuda").manual_seed(1)
        feats = torch.randn(n_img, 2, TEXT_LEN, 2048, device=dev, generator=g).to(torch.bfloat16)
        mask = torch.ones(n_img, 2, TEXT_LEN, dtype=torch.int32, device=dev)
        mask[:, 1, 8:] = 0  # empty prompt: BOS + padding (sample.py:38)
    feats, mask = parallel.broadcast_prompts(feats, mask, src=0, device=dev)
    mine = parallel.shard_range(n_img, rank, world)[0]
    cap_feats, cap_mask = feats[mine].contiguous(), mask[mine].contiguous()

    g = torch.Generator(device="cuda").manual_seed(100 + mine)
    z = torch.randn(1, 4, latent, latent, device=dev, generator=g).to(torch.bfloat16).repeat(2, 1, 1, 1)
    # sample.py:214-232: base_seqlen = (image_size // 16) ** 2 with the model's training image_size 1024
    kw = dict(cap_feats=cap_feats, cap_mask=cap_mask, cfg_scale=4.0, proportional_attn=True, base_seqlen=(1024 // 16) ** 2,
              scale_factor=wl["scale_factor"], scale_watershed=wl["scale_watershed"])
    transport = create_transport("Linear", "velocity", None, None, None)

    def run(nfe):
        fn = Sampler(transport).sample_ode(sampling_method="euler", num_steps=nfe + 1, time_shifting_factor=4)
        return fn(z, model.forward_with_cfg, **kw)

    if args.warmup > 0:
        run(args.warmup)
    torch.cuda.synchronize()
    eng = model._engine
    eng.profile_enable(args.profile_classes)
    # HIP start/stop events on the GEMM dispatches of the first --event-steps NFE of the timed region (every NFE has the
    # same launch mix; timing all of them costs ~2.5 ms per NFE of queue idle time, which would be charged to `value`)
    hd0 = model.dim // model.n_heads
    gemm_label, gemm_per_layer = gemm_kernel_label(_lib.load(), 2 * n_tokens, model.dim, model.ffn_hidden, model.n_kv_heads * hd0, n_tokens, hd0)
    gemm_launches_per_nfe = gemm_per_layer * model.n_layers + 2
    event_launches = -1 if args.event_steps <= 0 else max(1, int(round(args.event_steps * gemm_launches_per_nfe)))
    # ... taken from the middle of the region (the first launches after the barrier meet an idle chip in another power state)
    skip = gemm_launches_per_nfe * (args.steps // 2) if event_launches > 0 else 0
    eng.profile_set_window(0, skip, event_launches)
    eng.profile_reset()
    parallel.barrier()
    torch.cuda.synchronize()
    power = PowerSampler(local)
    with power:
        t0 = time.perf_counter()
        traj = run(args.steps)
        torch.cuda.synchronize()
        parallel.barrier()
        dt = time.perf_counter() - t0
    comm = parallel.comm_report(dt / args.steps * 1e3, dev)  # (after the timed region: its collectives are not part of `value`)
    dt = parallel.max_over_ranks(dt, dev)
    eng.profile_enable(False)
    assert eng.last_nfe() == args.steps
    gemm_prof = eng.profile_read(0)
    # further timed regions, same process, same bracket: not part of `value`
    repeats_ms = [dt / args.steps * 1e3]
    for _ in range(max(0, args.repeats - 1)):
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        parallel.barrier()
        repeats_ms.append(parallel.max_over_ranks(time.perf_counter() - t0, dev) / args.steps * 1e3)
    # untimed pass for the per-class breakdown (events around every launch)
    nb = min(4, args.steps)
    eng.profile_set_budget(0, -1)
    eng.profile_enable(7)
    eng.profile_reset()
    run(nb)
    torch.cuda.synchronize()
    eng.profile_enable(False)
    breakdown = {k: eng.profile_read(i)[0] / nb for i, k in enumerate(("gemm", "attention", "other"))}
    attn_ms_b, _, attn_fl_b = eng.profile_read(1)
    assert torch.isfinite(traj[-1].float()).all(), "non-finite latent"

    if rank == 0:
        hd = model.dim // model.n_heads
        nfe_flops = flops_per_nfe(dim=model.dim, n_layers=model.n_layers, n_heads=model.n_heads, n_kv_heads=model.n_kv_heads,
                                  ffn=model.ffn_hidden, cap_feat_dim=model.cap_feat_dim, n_tokens=n_tokens, text_len=TEXT_LEN, batch=2)
        gemm_ms, gemm_n, gemm_fl = gemm_prof
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        pmc_g, traffic_src = pmc_summary("pmc_gemm.json")
        traffic = None if pmc_g is None else float(pmc_g["hbm_bytes_per_launch"])
        pmc_a, attn_src = pmc_summary("pmc_attn.json")
        attn_tf = attn_fl_b / (attn_ms_b * 1e-3) / 1e12 if attn_ms_b > 0 else 0.0
        attn_launches = model.n_layers * nb
        # hd 72 on 32x32x16 MFMAs (csrc/attention_v4.hip): QK^T runs 5 k-steps = 80 of depth for 72 (the 8 pad slots carry the running max and
        # the text mask), P V writes 96 rows of O^T for 72 + the row of ones (row sum): executed / useful = (80 + 96) / (2 x 72)
        attn_exec = (80.0 + 96.0) / (2.0 * 72.0)
        rows_attn, dkv_attn = 2 * n_tokens, model.n_kv_heads * hd
        # bf16: q read + out written ([rows, d] each), k + v read ([rows, dkv] each), the text keys / values of the pair
        attn_alg_bytes = 2.0 * (2 * rows_attn * model.dim + 2 * rows_attn * dkv_attn + 2 * 2 * TEXT_LEN * dkv_attn)
        rp_ms, rp_src = rocprof_gemm_time_per_nfe()
        gemm_fl_per_nfe = gemm_fl / args.steps
        out = {
            "metric": "denoising-steps/s & latent-tokens/s, Next-DiT 2B %d^2 CFG" % res,
            "value": world * n_tokens * args.steps / dt,
            "unit": "latent-tokens/s",
            "n_gpus": 1 if share else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_repeats": repeats_ms,
            "box": None,  # filled below (needs the power report)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (wl["desc"] + ", text T=128, CFG=4 (cond+uncond B=2), proportional attention, "
                                    "flow-matching Euler ODE t_shift 4, 1 image per GPU"),
                       "images_per_gpu": 1, "nfe_timed": args.steps, "parallelism": f"images sharded x{world}, weights replicated"},
            "denoising_steps_per_s": world * args.steps / dt,
            "model_tflops_per_s_per_gpu": nfe_flops * args.steps / dt / 1e12,
            "mfma_roofline_frac_whole_step": nfe_flops * args.steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "roofline": {
                "bound": "mfma",
                "kernel": "all bf16 GEMM launches of the timed region - " + gemm_label,
                "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                "traffic_unit": "bytes/launch on the L2's FABRIC side (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE = TCC_EA read / write requests, "
                                "the guide's gfx950 correction applied): includes Infinity-Cache hits, so it is an upper bound on HBM bytes",
                "traffic_source": traffic_src,
                # A + W + C of every GEMM launch of one layer (QKV, O, W1|W3 + SwiGLU, W2), bf16, averaged over the launches
                "algorithmic_bytes_per_launch": algorithmic_gemm_bytes_per_launch(2 * n_tokens, model.dim, model.ffn_hidden,
                                                                                   model.n_kv_heads * hd),
                "traffic_over_algorithmic": (None if traffic is None else traffic / algorithmic_gemm_bytes_per_launch(
                    2 * n_tokens, model.dim, model.ffn_hidden, model.n_kv_heads * hd)),
                "launches": gemm_n, "avg_launch_ms": gemm_ms / max(gemm_n, 1),
                "event_bracketed_launches": (gemm_n if event_launches < 0 else min(gemm_n, event_launches)),
                "algorithmic_flops_per_launch": gemm_fl / max(gemm_n, 1),
                # the dispatch-attached event pairs include a few microseconds of packet processing per launch; the committed rocprofv3
                # kernel trace of the same command gives the pure kernel durations
                "rocprofv3_cross_check": (None if rp_ms is None else {
                    "gemm_ms_per_step": rp_ms, "achieved": gemm_fl_per_nfe / (rp_ms * 1e-3) / 1e12,
                    "frac": gemm_fl_per_nfe / (rp_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "source": rp_src,
                    "box": "builder's profiling box (a committed file), NOT this run's box"}),
            },
            # the kernel furthest below its roof: the fused self + text attention launch, 24 per NFE
            "roofline_attention": {
                "bound": "mfma",
                "kernel": "attn_fwd_kernel_v4<72>: flash_attn_varlen_func + the gated text SDPA of model.py:392-434 in one launch, q_norm + RoPE of the queries in its prologue",
                "achieved": attn_tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": attn_tf / MFMA_BF16_PEAK_TFLOPS,
                "achieved_is": "ALGORITHMIC flops 4 B H N (N + T) hd per launch / average launch duration from HIP events around every launch of an "
                               f"untimed pass of {nb} NFE right after the timed region (events around every launch cost the timed region 4 %)",
                "executed_over_algorithmic_mfma": attn_exec,
                "executed_mfma_frac": attn_tf * attn_exec / MFMA_BF16_PEAK_TFLOPS,
                "padding": "head_dim 72: QK^T at depth 80 (5 x 16; pad slots = running max + text mask), P V on 96 O^T rows (72 + the row of ones + zeros)",
                "launches": attn_launches, "avg_launch_ms": attn_ms_b / max(attn_launches, 1),
                "algorithmic_flops_per_launch": attn_fl_b / max(attn_launches, 1),
                "algorithmic_bytes_per_launch": attn_alg_bytes,
                "traffic": None if pmc_a is None else float(pmc_a["hbm_bytes_per_launch"]),
                "traffic_over_algorithmic": None if pmc_a is None else float(pmc_a["hbm_bytes_per_launch"]) / attn_alg_bytes,
                "mfma_duty_pmc": None if pmc_a is None else pmc_a.get("mfma_duty"),
                "traffic_unit": "bytes/launch on the L2's fabric side, as in `roofline`",
                "traffic_source": attn_src,
            },
            "kernel_time_ms_per_step": dict(breakdown, note=f"untimed pass of {nb} NFE with events around every launch"),
            "attention_tflops_per_s": attn_tf,
            "kernel_variants": {"attention": args.attn_variant or 4, "gemm": args.gemm_variant or 0,
                                # round 6: the persistent GEMM's A / W operands (and the four weights of every layer) in the row-pair-interleaved layout:
                                # whole 128-byte lines per LDS-DMA request (include/lumina_dit.h, lt_op_pair_layout); bit-identical results
                                "gemm_operand_layout": "row-major" if "pair_layout=0" in args.opt else "row-pair-interleaved"},
            "hip_graph_replays": eng.graph_replays(),
            "power": power.report(dt, nfe_flops * args.steps / 1e12),
            "parity_fixture": wl["parity"],
            "ode_stepping_parity": "euler (this run) and midpoint pinned to the reference's in-tree midpoint_solver (visual_anagrams/"
                                   "generate.py:212-219, tests/golden/solver_kat.npz); rk4 / dopri5 restate torchdiffeq (absent "
                                   "everywhere) and stay unpinned, DESIGN.md 6",
        }
        out["box"] = box_probe(_lib.load(), torch, out["power"])
        # evidence that the collective layer saw every rank: backend, RCCL version, an all-reduce of ones, each rank's own ms / step
        out["comm"] = dict(comm, ms_per_step_per_rank=comm["per_rank"], collective_in_timed_region=False,
                           note="value = images x tokens x NFE / max-over-ranks wall; no scaling curve has been measured on hardware "
                                "by the builder (1-GPU leases only) - the driver's SCALE file is the only source")
        del out["comm"]["per_rank"]
        if world > 1:
            assert comm["ranks_seen"] == world, comm
        if share:
            out.update(ranks=world, shared_device=True,
                       note="ranks share GPU 0 over gloo: exercises launcher / rank / shard / relay logic with the real engine; NOT a scaling "
                            "measurement and no evidence about RCCL or xGMI")
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg2":
            try:
                out["cpu_baseline"] = cpu_baseline(latent, n_tokens)
            except Exception as exc:  # the CPU leg is a reported baseline, not the product: its failure must not cost the measured line
                out["cpu_baseline"] = {"value": None, "unit": "latent-tokens/s", "cores": 0, "kind": "failed", "sample": "", "error": repr(exc)[:400]}
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
