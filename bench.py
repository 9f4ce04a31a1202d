#!/usr/bin/env python3
"""Benchmark of the batched per-window analysis path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (per-callback decisions + refractory pass ->
beat lists) over one batch of synthetic streams that is already resident in
HBM.  Default workload: the Spectrum strategy on 16,384 streams x 60 s at
44.1 kHz, 512-sample callbacks (1024-point analysis window, hop 512).  With N
GPUs every rank analyses its own 16,384 streams (weak scaling, no collective on
the data path); the reported value is the audio-seconds all ranks analysed per
wall-second, timed between barriers, maximum over ranks.

With --gpus N > 1 and no launcher environment (RANK / WORLD_SIZE unset) the
script starts its own N rank processes, one per GPU, before it touches the GPU
(with a wall limit: --launch-timeout); under `python -m torch.distributed.run`
it is a rank as before.  The process group only carries barriers, the
max-over-ranks of the timings and the per-rank report (gloo over loopback by
default, the other backend as fallback; waits bounded by --group-timeout).

Rank 0 prints one JSON line.  `value` / `roofline` time the launch that writes
every output of the batched entry (per callback the decision and the detection
value: 5 B); `value_decisions_only` / `roofline_decisions_only` time, over the
same number of steps, the launch that writes what the reference's strategies
return (decisions -> beats, 1 B per callback).  Both price the dominant kernel
against HBM bandwidth using HIP events recorded inside their timed region.
`ranks` lists, per rank, the GPU it ran on (index, name, PCI id / UUID), its
shard and its own timings; `backend` and `ranks_seen` say how they were
gathered.  `cpu_baseline` is
the CPU restatement of the reference (oracle/) timed on this host's cores on a
bounded sample of the same streams (N = 1 only), compared with the GPU's beat
lists stream by stream; `also` holds short
measurements of the other workloads (LPF, the extension analyses, one GPU's
share of the 131,072-track job through the chunked executor, live latency),
taken after and outside the headline's timed region.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (spec)
SR = 44100
SEED = 0x5EED

WORKLOADS = {
    # name: (kind, callback_len, streams, seconds, description)
    "spectrum_hop512": ("Spectrum", 512, 16384, 60,
                        "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, "
                        "1024-pt window hop 512 (512-sample callbacks), low-band max + threshold + refractory"),
    "spectrum_1024": ("Spectrum", 1024, 16384, 60,
                      "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, "
                      "1024-sample non-overlapping callbacks"),
    "lpf": ("LPF", 1024, 4096, 60,
            "LPF strategy batched: 4,096 synthetic click-track streams x 60 s @ 44.1 kHz, 1024-sample windows"),
    # callback lengths off the 64-sample grid (10 ms of audio at 44.1 kHz): other kernels behind the same entry
    "spectrum_441": ("Spectrum", 441, 16384, 60,
                     "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, 441-sample callbacks "
                     "against the 1024-sample ring"),
    # the callback size the reference's author reports on Linux ("mostly 500-540", spectrum.rs:40-42)
    "spectrum_529": ("Spectrum", 529, 16384, 60,
                     "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, 529-sample callbacks "
                     "against the 1024-sample ring"),
    # the usual low-latency buffer sizes: two and four callbacks per half-block of the same kernel
    "spectrum_256": ("Spectrum", 256, 16384, 60,
                     "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, 256-sample callbacks "
                     "against the 1024-sample ring"),
    "spectrum_128": ("Spectrum", 128, 16384, 60,
                     "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, 128-sample callbacks "
                     "against the 1024-sample ring"),
    "lpf_441": ("LPF", 441, 4096, 60,
                "LPF strategy batched: 4,096 synthetic click-track streams x 60 s @ 44.1 kHz, 441-sample windows"),
    # extension analysis (no reference counterpart): callback_len is the hop
    "onset_2048": ("Onset", 1024, 4096, 300,
                   "Extension mode: 4,096 synthetic 5-min tracks per GPU, 2048-pt Hann STFT hop 1024, spectral flux + "
                   "band energy + adaptive peak picking"),
    "onset_2048_power": ("Onset", 1024, 4096, 300,
                         "Extension mode: 4,096 synthetic 5-min tracks per GPU, 2048-pt Hann STFT hop 1024, flux of the "
                         "power spectrum (no square roots) + band energy + adaptive peak picking"),
    "biquad": ("Biquad", 1024, 4096, 60,
               "Extension mode: carried-state biquad envelope (120 Hz low-pass as a chunked scan), 4,096 synthetic "
               "click-track streams x 60 s @ 44.1 kHz, 1024-sample decision windows + refractory"),
    "onset_1024": ("Onset", 512, 4096, 300,
                   "Extension mode: 4,096 synthetic 5-min tracks per GPU, 1024-pt Hann STFT hop 512, spectral flux + "
                   "band energy + adaptive peak picking"),
    "onset_2048_hop512": ("Onset", 512, 4096, 300,
                          "Extension mode: 4,096 synthetic 5-min tracks per GPU, 2048-pt Hann STFT hop 512 (75 % "
                          "overlap), spectral flux + band energy + adaptive peak picking"),
    # one GPU's share of the 131,072-track job: more samples than HBM holds, through the chunked executor
    "onset_2048_chunked": ("OnsetChunked", 1024, 16384, 300,
                           "Extension mode, stream-sharded job of 131,072 synthetic 5-min tracks over 8 GPUs: 16,384 "
                           "tracks per GPU (433 GB of samples) in chunks of 4,096 tracks, chunk c+1 synthesised on a "
                           "side stream while chunk c is analysed; 2048-pt Hann STFT hop 1024, spectral flux + adaptive "
                           "peak picking"),
}
# FFT length of the onset workloads (callback_len is their hop)
ONSET_FFT = {"onset_2048": 2048, "onset_1024": 1024, "onset_2048_hop512": 2048, "onset_2048_chunked": 2048,
             "onset_2048_power": 2048}
METRIC = {"Spectrum": "1024-pt STFT", "LPF": "lowpass windows", "Biquad": "carried-state biquad"}


# --------------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher environment
# --------------------------------------------------------------------------------------------
def self_launch(n: int, wall_limit_s: float) -> int:
    """Starts the N ranks as child processes (this process never touches the GPU) and waits for them,
    at most `wall_limit_s` seconds: a rank that fails, or the limit, ends every other rank, and the
    launcher exits non-zero instead of hanging."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

    def end(pending):
        for q in pending:
            q.terminate()
        t_end = time.monotonic() + 10
        for q in pending:
            try:
                q.wait(max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()

    rc = 0
    pending = list(procs)
    deadline = time.monotonic() + wall_limit_s
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {procs.index(p)} exited with {code}; ending the other ranks", file=sys.stderr)
                end(pending)   # a rank failed: the others would wait at a barrier until its timeout
                pending = []
                break
        if pending and time.monotonic() > deadline:
            print(f"bench.py: {len(pending)} of {n} ranks still running after {wall_limit_s:.0f} s "
                  "(--launch-timeout); ending them", file=sys.stderr)
            end(pending)
            return 124
        time.sleep(0.05)
    return rc


def init_group(args, rank: int, world: int, device=None):
    """The process group of a multi-rank run.  The data path has no collective (streams are independent:
    one AnalysisState and one ring per detector, src/strategies/mod.rs:39-56 of the reference), so the group
    only carries the barriers around the timed region, the max-over-ranks of the timings and the per-rank
    report: gloo over loopback by default; whichever of gloo / nccl is not the first choice is the fallback
    when the first fails to initialise.  Every wait is bounded by --group-timeout.
    Returns (torch.distributed, backend used, device for reduction tensors, notes)."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost") and os.path.isdir("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # the host name of a container may not resolve
    timeout = datetime.timedelta(seconds=args.group_timeout)
    order = [args.backend] + [b for b in ("gloo", "nccl") if b != args.backend]
    notes = []
    for backend in order:
        try:
            if backend == "nccl":
                if device is None:
                    raise RuntimeError("nccl (RCCL) needs a GPU per rank; this rank has none")
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=timeout)
                t = torch.ones(1, device=device)
                dist.all_reduce(t)                          # RCCL builds its communicator on first use
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
                t = torch.ones(1)
                dist.all_reduce(t)
            if int(t[0]) != world:
                raise RuntimeError(f"first all-reduce over {backend} saw {int(t[0])} of {world} ranks")
            return dist, backend, (device if backend == "nccl" else torch.device("cpu")), notes
        except Exception as e:  # noqa: BLE001 - whatever the backend raises, the other one is tried
            notes.append(f"{backend}: {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}")
            print(f"bench.py: rank {rank}: process group over {backend} failed ({notes[-1]})", file=sys.stderr)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
    raise SystemExit("bench.py: no process-group backend could be initialised: " + "; ".join(notes))


def gather_ranks(dist, world: int, mine: dict):
    """Every rank's report, in rank order (rank 0 prints them as `ranks`)."""
    if dist is None:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def rehearse_launch(args) -> None:
    """No GPU work: the ranks join the group and exchange their reports as a real run does."""
    from beat_detector_amd import sharding
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    dist, backend, notes = None, None, []
    if world > 1:
        dist, backend, _, notes = init_group(args, rank, world, None)
    if args.rehearse_sleep:
        time.sleep(args.rehearse_sleep)
    streams = WORKLOADS[args.workload][2]
    first, last = sharding.shard_range(world * streams, rank, world)
    ranks = gather_ranks(dist, world, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                                       "pid": os.getpid(), "host": socket.gethostname(), "device_index": None,
                                       "device_name": None, "device_id": None, "shard": [first, last],
                                       "kernel_ms": None})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"rehearsal": "launch", "backend": backend, "backend_notes": notes, "ranks_seen": len(ranks), "ranks": ranks,
                          "shards": [r["shard"] for r in ranks]}), flush=True)


class Ctx:
    """Process-group plumbing shared by the measurements."""

    def __init__(self, args):
        import torch
        import beat_detector_amd as bd
        self.torch, self.bd, self.args = torch, bd, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != max(args.gpus, 1):
            raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU path)")
        index = 0 if args.share_device else self.local_rank
        if index >= torch.cuda.device_count():
            raise SystemExit(f"rank {self.rank}: local rank {self.local_rank} has no GPU of its own "
                             f"({torch.cuda.device_count()} visible); --share-device runs every rank on GPU 0")
        torch.cuda.set_device(index)
        self.dev = torch.device("cuda", index)
        self.dist, self.backend, self.backend_notes = None, None, []
        self.red_dev = torch.device("cpu")
        if self.world > 1:
            self.dist, self.backend, self.red_dev, self.backend_notes = init_group(args, self.rank, self.world, self.dev)
        self.lib = bd.lib()
        self.st = torch.cuda.current_stream().cuda_stream

    def report(self, shard, kernel_ms, ms_per_step):
        """What this rank contributes to the `ranks` list: which physical GPU it ran on, its shard, its timings."""
        torch = self.torch
        props = torch.cuda.get_device_properties(self.dev)
        ident = None
        if hasattr(props, "pci_bus_id"):
            ident = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
        uuid = str(getattr(props, "uuid", "")) or None
        try:
            name = self.bd.device_name(self.dev.index)[0]      # bd_device_name: what the library itself runs on
        except Exception:  # noqa: BLE001
            name = props.name
        return {"rank": self.rank, "local_rank": self.local_rank, "pid": os.getpid(), "host": socket.gethostname(),
                "device_index": self.dev.index, "device_name": name, "device_id": ident, "device_uuid": uuid,
                "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"),
                "shard": [int(shard[0]), int(shard[1])], "kernel_ms": kernel_ms, "ms_per_step": ms_per_step}

    def ranks(self, shard, kernel_ms, ms_per_step):
        """Gathers every rank's report; two ranks on one physical GPU are an error unless --share-device."""
        reports = gather_ranks(self.dist, self.world, self.report(shard, kernel_ms, ms_per_step))
        seen = {}
        for r in reports:
            key = (r["host"], r.get("visible_devices"), r["device_uuid"] or r["device_id"] or r["device_index"])
            if key in seen and not self.args.share_device:
                raise SystemExit(f"bench.py: ranks {seen[key]} and {r['rank']} both ran on GPU {key[2]} of {key[0]}: "
                                 "a multi-GPU measurement needs one GPU per rank (--share-device is for rehearsals)")
            seen[key] = r["rank"]
        return reports

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, *xs):
        if self.dist is None:
            return xs
        t = self.torch.tensor(list(xs), dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    def sum_over_ranks(self, x: int) -> int:
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.int64, device=self.red_dev)
        self.dist.all_reduce(t)
        return int(t[0])

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.bd_last_error().decode())

    def shard(self, streams):
        """[first, last) of this rank's streams: weak scaling, `streams` per GPU."""
        from beat_detector_amd import sharding
        return sharding.shard_range(self.world * streams, self.rank, self.world)

    def first_stream(self, streams):
        return self.shard(streams)[0]


# --------------------------------------------------------------------------------------------
# a workload whose samples are resident in HBM
# --------------------------------------------------------------------------------------------
def run_resident(ctx, workload, steps, warmup, streams=0, seconds=0, outputs="both"):
    """Measures one workload; returns (result dict, state kept for the CPU baseline).

    The two strategies are timed twice over the same `steps`, each between barriers: once with every
    output of the batched entry written (per callback the decision, 1 B, and the detection value, 4 B:
    the headline, `value` / `roofline`), once with the decisions alone (`value_decisions_only` /
    `roofline_decisions_only`: what the reference's strategies return, Strategy::is_beat -> Option<BeatInfo>,
    src/lib.rs:72-78, has no value in it).  `outputs` = "values" or "decisions" runs one of the two only
    (profiling runs: every launch of the dominant kernel is then of one kind)."""
    import ctypes
    torch, bd, L_ = ctx.torch, ctx.bd, ctx.lib
    kind_name, L, d_streams, d_seconds, desc = WORKLOADS[workload]
    streams = streams or d_streams
    seconds = seconds or d_seconds
    onset = kind_name == "Onset"
    biquad = kind_name == "Biquad"
    kind = None if onset else bd.StrategyKind["LPF" if biquad else kind_name]
    sr, dev, st = SR, ctx.dev, ctx.st
    n = sr * seconds
    stride = (n + 63) & ~63  # rows start on 128-byte boundaries
    # weak scaling: `streams` per GPU, rank r owns the r-th contiguous shard of world * streams
    shard = ctx.shard(streams)
    first_stream = shard[0]

    samples = torch.empty((streams, stride), dtype=torch.int16, device=dev)
    bd.synth_fill(samples, n_samples=n, first_stream=first_stream, seed=SEED, sampling_rate=sr)
    nfft = ONSET_FFT.get(workload, 0)
    if onset:
        oparams = bd.OnsetParams.default(fft_len=nfft, hop=L, flux_power=1 if workload.endswith("_power") else 0)
        n_cb = bd.onset_frames(n, nfft, L)               # frames per track
        max_beats = max(1, (n * 1000 // sr) // max(oparams.min_gap_ms, 1) + 1)
    else:
        n_cb = n // L
        max_beats = max(1, bd.batch_max_beats(kind, sr, n, L))
    # output rows start on 128-byte boundaries so that every store is a full line
    out_stride = (n_cb + 127) & ~127
    values = torch.empty((streams, out_stride), dtype=torch.float32, device=dev)
    flags = torch.empty((streams, out_stride), dtype=torch.float32 if onset else torch.uint8, device=dev)
    count = torch.zeros((streams,), dtype=torch.int32, device=dev)
    idx = torch.zeros((streams, max_beats), dtype=torch.int32, device=dev)
    ms = torch.zeros((streams, max_beats), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    two_modes = not (onset or biquad)        # the value rows are optional for the two strategies only
    if onset:
        def stage1(with_values=True):  # `values` holds the flux rows, `flags` the band-energy rows
            ctx.check(L_.bd_onset_stft(sr, ctypes.byref(oparams), samples.data_ptr(), streams, n, stride,
                                       values.data_ptr(), flags.data_ptr(), out_stride, st))

        def stage2():
            ctx.check(L_.bd_onset_pick(sr, ctypes.byref(oparams), values.data_ptr(), streams, n_cb, out_stride, None,
                                       count.data_ptr(), idx.data_ptr(), ms.data_ptr(), max_beats, st))
    else:
        if biquad:
            bq_params = bd.BiquadParams.default(window_len=L)
            bq_design = bd.biquad_design(sr, bq_params.cutoff_hz, bq_params.q_factor)

            def stage1(with_values=True):
                ctx.check(L_.bd_biquad_envelope(sr, ctypes.byref(bq_params), ctypes.byref(bq_design), samples.data_ptr(),
                                                streams, n, stride, values.data_ptr(), flags.data_ptr(), out_stride, st))
        else:
            def stage1(with_values=True):
                ctx.check(L_.bd_batch_candidates(kind.value, sr, None, samples.data_ptr(), streams, n, stride, L,
                                                 values.data_ptr() if with_values else None, flags.data_ptr(), out_stride, st))

        def stage2():
            ctx.check(L_.bd_refractory(sr, None, L, flags.data_ptr(), out_stride, streams, n_cb, count.data_ptr(),
                                       idx.data_ptr(), ms.data_ptr(), max_beats, st))

    def timed(with_values):
        """`warmup` untimed steps, then exactly `steps` steps between barriers; (wall seconds, dominant
        kernel ms, second pass ms) with HIP events on the launch stream inside the timed region."""
        for _ in range(warmup):
            stage1(with_values)
            stage2()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        ctx.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            stage1(with_values)           # dominant kernel
            ev[k][1].record()
            stage2()                      # refractory pass / peak picking
            ev[k][2].record()
        ctx.barrier()
        elapsed = time.perf_counter() - t0
        if os.environ.get("BENCH_VERBOSE"):
            print("per-step kernel ms (values written: %s):" % with_values,
                  [round(e[0].elapsed_time(e[1]), 3) for e in ev], file=sys.stderr)
        k1 = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
        k2 = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
        elapsed_max, k1_max = ctx.max_over_ranks(elapsed, k1)
        return {"elapsed": elapsed_max, "k1_ms": k1_max, "k2_ms": k2, "own_k1_ms": k1, "own_ms_per_step": elapsed / steps * 1e3}

    stage1(outputs != "decisions")
    launched = bd.last_kernel_name()      # the kernel the entry point routed this workload to
    stage2()
    main_with_values = outputs != "decisions" or not two_modes
    t_main = timed(main_with_values)
    t_dec = timed(False) if (two_modes and outputs == "both") else None
    if two_modes and outputs == "both":
        stage1(True)                      # the value rows feed the comparison with the CPU baseline
    torch.cuda.synchronize()
    kernel_name = launched or {"Spectrum": "bd_spectrum_band_kernel", "LPF": "bd_lpf_window_kernel",
                               "Onset": "bd_stft_flux_kernel", "Biquad": "bd_biquad_env_kernel"}[kind_name]
    total_beats = ctx.sum_over_ranks(int(count.sum().item()))
    world = ctx.world

    # algorithmic bytes of one launch of the dominant kernel: every sample that belongs to a callback
    # read once (2 B) + per callback the decision (1 B) and, where the launch writes it, the value (4 B)
    def alg_bytes_of(with_values):
        if onset:  # every sample of a complete frame once + flux and band energy (8 B) per frame
            return streams * (((n_cb - 1) * L + nfft) * 2 + n_cb * 8) if n_cb else 0
        return streams * (n_cb * L * 2 + n_cb * (5 if with_values else 1))

    def roofline_of(t, with_values):
        alg = alg_bytes_of(with_values)
        achieved = alg / (t["k1_ms"] * 1e-3) / 1e9
        traffic, traffic_source = _pmc_traffic(workload, streams, seconds, with_values or not two_modes)
        return {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg, "kernel_ms": t["k1_ms"], "refractory_ms": t["k2_ms"],
                "outputs_written": ("decisions + detection values" if with_values else "decisions") if two_modes
                                   else "every output of the entry point"}

    result = {
        "metric": "audio-sec analysed/sec (node), " + ("%d-pt STFT" % nfft if onset else METRIC[kind_name]),
        "value": world * streams * (n / sr) * steps / t_main["elapsed"],
        "unit": "audio-sec/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": t_main["elapsed"] / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc,
            "strategy": kind_name,
            "streams_per_gpu": streams,
            "seconds_per_stream": seconds,
            "sampling_rate": sr,
            "callback_len": L,
            "callbacks_per_stream": n_cb,
            "seed": SEED,
            "parallelism": f"stream-sharded x{world}, no collectives",
            "beats_found": total_beats,
            "outputs": ("decisions + detection values + beat lists" if main_with_values
                        else "decisions + beat lists (what the reference's strategies return); detection values not written"),
        },
        "roofline": roofline_of(t_main, main_with_values),
    }
    if t_dec is not None:
        result["value_decisions_only"] = world * streams * (n / sr) * steps / t_dec["elapsed"]
        result["ms_per_step_decisions_only"] = t_dec["elapsed"] / steps * 1e3
        result["roofline_decisions_only"] = roofline_of(t_dec, False)
    if onset:
        result["roofline"]["note"] = ("this kernel is limited by f32 VALU issue (a 2048-point FFT per 1024 new samples), "
                                      "not by HBM; the HBM fraction is reported for comparison with the other workloads")
        result["roofline"]["valu"] = _valu_roofline(streams, n_cb, nfft, t_main["k1_ms"])
    # which GPU every rank ran on, its shard and its own timings (the driver can check N distinct devices)
    result["backend"] = ctx.backend
    result["ranks"] = ctx.ranks(shard, t_main["own_k1_ms"], t_main["own_ms_per_step"])
    result["ranks_seen"] = len(result["ranks"])
    if ctx.backend_notes:
        result["backend_notes"] = ctx.backend_notes
    state = dict(flux_power=1 if workload.endswith("_power") else 0, kind_name=kind_name, samples=samples, values=values, flags=flags, count=count, idx=idx, ms=ms, n=n, L=L,
                 sr=sr, nfft=nfft, streams=streams, out_stride=out_stride, max_beats=max_beats,
                 has_values=main_with_values or t_dec is not None)
    return result, state


def _valu_roofline(streams, n_frames, nfft, k_ms):
    # a full FFT per frame is bounded by f32 vector throughput as well: 2.5 N log2 N flops per real FFT
    # against the packed-FMA vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 x 2 at 2.4 GHz)
    flops = streams * n_frames * 2.5 * nfft * (nfft.bit_length() - 1)
    peak_tflops = 256 * 4 * 16 * 2 * 2 * 2.4e9 / 1e12
    return {"achieved": flops / (k_ms * 1e-3) / 1e12, "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": flops / (k_ms * 1e-3) / 1e12 / peak_tflops, "flops_per_frame": 2.5 * nfft * (nfft.bit_length() - 1)}


def _pmc_traffic(workload, streams, seconds, with_values=True):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  The figure is not
    measured in this run: it is replayed from the committed rocprofv3 PMC passes of this same command
    (`--outputs values` for the launch that writes the detection values, `--outputs decisions` for the other)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get(workload if with_values else workload + "_decisions")
        if e and e.get("streams") == streams and e.get("seconds") == seconds:
            return e.get("hbm_bytes_per_launch"), ("committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                                   "(%s), not measured in this run" % e.get("source", "profiles/"))
    except Exception:
        pass
    return None, None


# --------------------------------------------------------------------------------------------
# one GPU's share of the 131,072-track job: chunked executor, samples never resident at once
# --------------------------------------------------------------------------------------------
def run_chunked_onset(ctx, steps, warmup, tracks=0, seconds=0, chunk=0):
    import ctypes
    torch, bd = ctx.torch, ctx.bd
    workload = "onset_2048_chunked"
    _, hop, d_tracks, d_seconds, desc = WORKLOADS[workload]
    tracks = tracks or d_tracks
    seconds = seconds or d_seconds
    chunk = min(chunk or 4096, tracks)
    nfft = ONSET_FFT[workload]
    n = SR * seconds
    p = bd.OnsetParams.default(fft_len=nfft, hop=hop)
    n_frames = bd.onset_frames(n, nfft, hop)
    max_onsets = max(1, (n * 1000 // SR) // max(p.min_gap_ms, 1) + 1)
    first = ctx.first_stream(tracks)
    ex = bd.ChunkedExecutor(chunk, n)
    count = torch.zeros((tracks,), dtype=torch.int32, device=ctx.dev)
    frame = torch.zeros((tracks, max_onsets), dtype=torch.int32, device=ctx.dev)
    ms = torch.zeros((tracks, max_onsets), dtype=torch.int32, device=ctx.dev)

    def step():
        ex.onset(SR, tracks, count, frame, ms, params=p, first_stream=first, seed=SEED)

    for _ in range(max(warmup, 1)):
        step()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    main_ms, post_ms = ex.last_timing()            # the chunks of the last timed pass
    own_k_ms = k_ms = sum(main_ms) / max(len(main_ms), 1)
    own_ms_per_step = elapsed / steps * 1e3
    total = ctx.sum_over_ranks(int(count.sum().item()))
    elapsed, k_ms = ctx.max_over_ranks(elapsed, k_ms)
    # The same passes with a source that enqueues nothing (the two chunk buffers keep the samples of the
    # chunks synthesised last): the analysis of every chunk and the executor's hand-over events without
    # the synthetic source's cost.  The onsets of these passes are those of the stale samples and are not
    # looked at.
    scratch_count = torch.zeros_like(count)

    def null_step():
        ex.onset(SR, tracks, scratch_count, frame, ms, params=p, source=lambda *a: 0)

    null_step()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        null_step()
    ctx.barrier()
    elapsed_null = time.perf_counter() - t0
    null_main_ms, _ = ex.last_timing()
    k_ms_null = sum(null_main_ms) / max(len(null_main_ms), 1)
    elapsed_null, k_ms_null = ctx.max_over_ranks(elapsed_null, k_ms_null)
    dev_bytes = ex.device_bytes()
    n_chunks = (tracks + chunk - 1) // chunk
    alg_bytes = chunk * (((n_frames - 1) * hop + nfft) * 2 + n_frames * 4) if n_frames else 0   # per launch = per chunk
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
    result = {
        "workload": workload,
        "metric": "audio-sec analysed/sec (node), %d-pt STFT" % nfft,
        "value": ctx.world * tracks * (n / SR) * steps / elapsed,
        "unit": "audio-sec/s",
        "n_gpus": ctx.world,
        "steps": steps,
        "ms_per_step": elapsed / steps * 1e3,
        "value_excluding_synthesis": ctx.world * tracks * (n / SR) * steps / elapsed_null,
        "ms_per_step_excluding_synthesis": elapsed_null / steps * 1e3,
        "scaling": "weak",
        "config": {"workload": desc, "tracks_per_gpu": tracks, "seconds_per_track": seconds, "chunk_tracks": chunk,
                   "chunks_per_pass": n_chunks, "frames_per_track": n_frames, "device_bytes_held": dev_bytes,
                   "samples_bytes_per_pass": tracks * n * 2, "onsets_found": total, "seed": SEED,
                   "timed_region": "whole pass: on-device synthesis of every chunk (side stream, overlapped) + STFT/flux + "
                                   "peak picking; the samples of a pass never exist at once",
                   "timed_region_excluding_synthesis": "the same pass with a source that enqueues nothing (the chunk "
                                                       "buffers keep the last synthesised chunks): analysis of every "
                                                       "chunk + hand-over events, no generator"},
        "roofline": {"bound": "hbm", "kernel": "bd_stft_flux_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms,
                     "peak_picking_ms": sum(post_ms) / max(len(post_ms), 1),
                     "kernel_ms_per_chunk": [round(x, 3) for x in main_ms],
                     "kernel_ms_excluding_synthesis": k_ms_null,
                     "frac_excluding_synthesis": (alg_bytes / (k_ms_null * 1e-3) / 1e9 / HBM_PEAK_GBPS) if k_ms_null else None,
                     "note": "VALU-issue bound (a 2048-point f32 FFT per 1024 new samples); launches overlap the synthesis "
                             "of the next chunk",
                     "valu": _valu_roofline(chunk, n_frames, nfft, k_ms) if k_ms else None},
    }
    result["backend"] = ctx.backend
    result["ranks"] = ctx.ranks(ctx.shard(tracks), own_k_ms, own_ms_per_step)
    result["ranks_seen"] = len(result["ranks"])
    ex.close()
    del count, frame, ms, scratch_count
    return result


# --------------------------------------------------------------------------------------------
# live path: 256 concurrent streams, one tick = one 1024-sample callback per stream
# --------------------------------------------------------------------------------------------
def run_live(ctx, ticks=400, streams=256, L=1024):
    import gc
    import numpy as np
    bd = ctx.bd
    out = {"workload": "live_256", "streams": streams, "callback_len": L, "ticks": ticks,
           "tick_period_us": L / SR * 1e6,
           "measured": "submit -> result on the host, per tick (pinned ring, one HIP graph replay per tick)"}
    x = bd.synth_fill_host(streams, L * 64, seed=11)
    wins = [np.ascontiguousarray(x[:, i * L:(i + 1) * L]) for i in range(64)]
    # the same with callbacks of 500-540 samples, a new length on most ticks (what the reference's
    # author reports for Linux, spectrum.rs:40-42)
    rng = np.random.default_rng(1)
    var_len = rng.integers(500, 541, ticks)
    # and with every stream drawing its own 500-540 samples per tick (each stream is a device of its own in
    # the reference, record.rs:93-95: concurrent streams do not deliver equal lengths): bd_live_submit_ragged
    ragged_len = rng.integers(500, 541, (ticks, streams)).astype(np.uint32)
    ragged_blk = [np.ascontiguousarray(x[:, 100 * i:100 * i + 540]) for i in range(8)]
    for kind in (bd.StrategyKind.Spectrum, bd.StrategyKind.LPF):
        for label, lengths in ((kind.name, None), (kind.name + "_500_540", var_len), (kind.name + "_ragged_500_540", "ragged")):
            ragged = isinstance(lengths, str)
            live = bd.Live(kind, SR, streams, max_callback_len=L, depth=4)
            if ragged:
                blocks = ragged_blk
            else:
                blocks = wins if lengths is None else [np.ascontiguousarray(x[:, 100 * (t % 8):100 * (t % 8) + int(n)])
                                                       for t, n in enumerate(lengths)]
            lat = []
            gc.collect()
            gc.disable()   # a collection of the interpreter's heap inside a tick is not the path's latency
            for t in range(ticks):                     # every tick counts: the first ones included
                blk = blocks[t % len(blocks)]
                t0 = time.perf_counter()
                tick = live.submit_ragged(blk, ragged_len[t]) if ragged else live.submit(blk)
                live.poll(tick)
                lat.append((time.perf_counter() - t0) * 1e6)
            gc.enable()
            live.close()
            lat = np.array(lat)
            out[label] = {"p50_us": float(np.percentile(lat, 50)), "p99_us": float(np.percentile(lat, 99)),
                          "max_us": float(lat.max()), "first_tick_us": float(lat[0])}
    return out


# --------------------------------------------------------------------------------------------
# CPU baseline: the oracle on this host's cores
# --------------------------------------------------------------------------------------------
def _host_cpus():
    """(threads to use, description of the host's CPUs)."""
    total = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:
        allowed = total
    use = allowed
    share = os.environ.get("OMP_NUM_THREADS", "")
    note = "all CPUs this process may run on"
    if share.isdigit() and 0 < int(share) < use:
        # a job on a shared machine gets a CPU share; the harness publishes it in OMP_NUM_THREADS
        use = int(share)
        note = f"the job's CPU share (OMP_NUM_THREADS={share}) of a host with {total} logical CPUs"
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return use, {"cpu_model": model, "cores_total": total, "cores_allowed": allowed, "cores_used": use, "cores_note": note}


def cpu_baseline(ctx, state, n_streams):
    """The CPU restatement of the reference (oracle/, a port: the Rust original
    cannot be built here) on the first `n_streams` streams of the same batch,
    compared with the GPU's results stream by stream."""
    import numpy as np
    kind_name = state["kind_name"]
    if kind_name == "Onset":
        return cpu_baseline_onset(state, min(n_streams, 128))
    if kind_name == "Biquad":
        return cpu_baseline_biquad(state, min(n_streams, 512))
    from oracle import bd_oracle
    okind = bd_oracle.KIND_SPECTRUM if kind_name == "Spectrum" else bd_oracle.KIND_LPF
    n, L, sr = state["n"], state["L"], state["sr"]
    n_streams = min(n_streams, state["streams"])
    host = state["samples"][:n_streams, :n].cpu().numpy()
    cores, info = _host_cpus()
    max_beats = state["max_beats"]
    t0 = time.perf_counter()
    total, counts, cpu_idx, cpu_ms = bd_oracle.run_batch_lists(okind, sr, host, L, max_beats, n_threads=cores)
    dt = time.perf_counter() - t0
    one = max(1, n_streams // (cores * 2))
    t1 = time.perf_counter()
    bd_oracle.run_batch(okind, sr, host[:one], L, n_threads=1)
    dt1 = time.perf_counter() - t1
    # stream by stream: the beat lists themselves (callback index and relative_ms of every beat), not their lengths
    gpu_counts = state["count"][:n_streams].cpu().numpy().astype(np.int64)
    gpu_idx = state["idx"][:n_streams].cpu().numpy().astype(np.int64)
    gpu_ms = state["ms"][:n_streams].cpu().numpy().astype(np.int64)
    used = np.arange(max_beats)[None, :] < np.minimum(counts.astype(np.int64), max_beats)[:, None]
    same = (gpu_counts == counts.astype(np.int64)) & \
           ((gpu_idx == cpu_idx.astype(np.int64)) | ~used).all(axis=1) & ((gpu_ms == cpu_ms.astype(np.int64)) | ~used).all(axis=1)
    differing = np.nonzero(~same)[0]
    out = {
        "value": n_streams * (n / sr) / dt,
        "unit": "audio-sec/s",
        "cores": cores,
        "kind": "port",
        "label": "CPU restatement of the reference (oracle/), not the Rust original",
        "sample": f"first {n_streams} streams x {n // sr} s of the same batch, {cores} threads "
                  f"(one contiguous stream shard each), {dt:.1f} s of wall time; beats found {total}",
        "cpu_events": int(total),
        "gpu_events_same_sample": int(gpu_counts.sum()),
        "streams_compared": int(n_streams),
        "compared": "per stream, the whole beat list: callback index and relative_ms of every beat, bit-exact",
        "streams_differing": int(len(differing)),
        "streams_differing_list": [int(s) for s in differing[:64]],
        "value_1core": one * (n / sr) / dt1,
        "sample_1core": f"first {one} streams, 1 thread, {dt1:.1f} s",
    }
    out.update(info)
    if kind_name == "Spectrum" and state.get("has_values"):
        out.update(_threshold_adjacent(ctx, state, host, n_streams, bd_oracle))
    return out


def _threshold_adjacent(ctx, state, host, n_streams, bd_oracle, rel=1e-4, threshold=2.1e6):
    """The callbacks of the compared streams whose reference value lies within `rel` of the decision
    threshold: the only ones whose decision may differ between the f32 FFT of the CPU restatement and the
    pruned DFT of the kernel.  Candidates are taken from the GPU values with twice the margin (values agree
    to `rel`), then the oracle's own value decides membership."""
    import numpy as np
    torch = ctx.torch
    n, L = state["n"], state["L"]
    n_cb = n // L
    vals = state["values"][:n_streams, :n_cb]
    near = torch.nonzero((vals - threshold).abs() <= 2 * rel * threshold).cpu().numpy()
    listed = []
    for s, j in near:
        end = (int(j) + 1) * L
        ring = np.zeros(1024, dtype=np.float32)          # ring pre-filled with zeros, spectrum.rs:50-51
        lo = max(0, end - 1024)
        ring[1024 - (end - lo):] = host[s, lo:end]
        ref, _ = bd_oracle.spectrum_band_max(ring, state["sr"])
        if abs(ref - threshold) <= rel * threshold:
            g = float(vals[s, j])
            listed.append({"stream": int(s), "callback": int(j), "oracle_value": ref, "gpu_value": g,
                           "oracle_decision_value_above": bool(ref > threshold), "gpu_decision_value_above": bool(g > threshold)})
    return {"threshold_adjacent_callbacks": listed[:64], "threshold_adjacent_count": len(listed),
            "threshold_adjacent_rule": f"|oracle value - {threshold:g}| <= {rel:g} * {threshold:g}, over "
                                       f"{n_streams} streams x {n_cb} callbacks"}


def cpu_baseline_biquad(state, n_streams):
    """The CPU definition of the carried-state biquad envelope (oracle/bd_oracle_ext.c)
    plus the refractory replay, one stream per task."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import bd_oracle, bd_oracle_ext as ox
    n, L, sr = state["n"], state["L"], state["sr"]
    host = state["samples"][:n_streams, :n].cpu().numpy()
    cores, info = _host_cpus()
    p = ox.biquad_params(window_len=L)
    d = ox.biquad_design(sr, p.cutoff_hz, p.q_factor)
    ol = bd_oracle.lib()

    def one(row):
        _, flags = ox.biquad_envelope(row, sr, p, d)
        st = bd_oracle.State()
        ol.bdo_state_new(st, sr)
        beats = 0
        for f in flags:
            ol.bdo_state_update_time(st, L)
            if f and ol.bdo_last_beat_beyond_threshold(st, 400):
                ol.bdo_state_mark_beat(st)
                beats += 1
        return beats

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        per_stream = list(ex.map(one, host))
    dt = time.perf_counter() - t0
    return _ext_baseline(state, n_streams, per_stream, dt, cores, info,
                         "the definition of the extension analysis; refractory replay in Python")


def cpu_baseline_onset(state, n_streams):
    """The CPU definition of the extension analysis (oracle/bd_oracle_ext.c) on the
    first `n_streams` tracks; ctypes releases the GIL, one track per task."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import bd_oracle_ext as ox
    n, hop, sr, nfft = state["n"], state["L"], state["sr"], state["nfft"]
    host = state["samples"][:n_streams, :n].cpu().numpy()
    cores, info = _host_cpus()
    p = ox.default_params(fft_len=nfft, hop=hop, flux_power=state.get("flux_power", 0))

    def one(row):
        flux, _ = ox.onset_stft(row, sr, p)
        return len(ox.onset_pick(flux, sr, p)[1])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        per_stream = list(ex.map(one, host))
    dt = time.perf_counter() - t0
    return _ext_baseline(state, n_streams, per_stream, dt, cores, info,
                         "scalar radix-2 FFT: the definition of the analysis, not a tuned CPU FFT")


def _ext_baseline(state, n_streams, per_stream, dt, cores, info, what):
    import numpy as np
    n, sr = state["n"], state["sr"]
    gpu = state["count"][:n_streams].cpu().numpy().astype(np.int64)
    differing = np.nonzero(gpu != np.asarray(per_stream, dtype=np.int64))[0]
    out = {
        "value": n_streams * (n / sr) / dt,
        "unit": "audio-sec/s",
        "cores": cores,
        "kind": "port",
        "label": "CPU definition of an extension analysis (no reference counterpart)",
        "sample": f"first {n_streams} streams x {n // sr} s of the same batch, {cores} threads, {dt:.1f} s of wall "
                  f"time; events found {sum(per_stream)} ({what})",
        "cpu_events": int(sum(per_stream)),
        "gpu_events_same_sample": int(gpu.sum()),
        "streams_compared": int(n_streams),
        "streams_differing": int(len(differing)),
        "streams_differing_list": [int(s) for s in differing[:64]],
    }
    out.update(info)
    return out


# --------------------------------------------------------------------------------------------
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="spectrum_hop512", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU (testing only)")
    ap.add_argument("--seconds", type=int, default=0, help="override stream length in seconds (testing only)")
    ap.add_argument("--chunk-tracks", type=int, default=0, help="chunked workload: tracks per chunk (default 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-streams", type=int, default=1024, help="streams in the CPU baseline sample")
    ap.add_argument("--also", default="auto", choices=["auto", "all", "config4", "none"],
                    help="extra short measurements attached as `also` (auto: all at N = 1 with the default workload, "
                         "the chunked 131,072-track share at N > 1)")
    ap.add_argument("--also-tracks", type=int, default=0, help="tracks per GPU of the chunked `also` entry (rehearsals)")
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"],
                    help="process-group backend for the barriers, the timing reduction and the per-rank report (the "
                         "data path has no collective); the other one is the fallback if this one fails to initialise")
    ap.add_argument("--group-timeout", type=float, default=600.0,
                    help="seconds any wait on the process group may take before the rank gives up")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="self-launched ranks (--gpus N without a launcher): wall limit in seconds, after which the "
                         "ranks are ended and the launcher exits with 124")
    ap.add_argument("--outputs", default="both", choices=["both", "values", "decisions"],
                    help="the two strategies: time the launch that writes detection values, the one that writes "
                         "decisions only, or both (default; profiling runs pick one)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a box with one GPU)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU work: every rank joins a gloo group, the ranks' numbers are summed and rank 0 prints "
                         "them (checks the launcher on a machine without a GPU)")
    ap.add_argument("--rehearse-sleep", type=float, default=0.0, help=argparse.SUPPRESS)   # tests: a rank that hangs
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, args.launch_timeout))        # nothing above touched the GPU

    if args.rehearse_launch:
        return rehearse_launch(args)
    ctx = Ctx(args)
    torch = ctx.torch
    chunked = WORKLOADS[args.workload][0] == "OnsetChunked"
    if chunked:
        r = run_chunked_onset(ctx, args.steps, args.warmup, args.streams, args.seconds, args.chunk_tracks)
        r.update({"warmup": args.warmup, "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic"})
        r.pop("workload")
        result, state = r, None
    else:
        result, state = run_resident(ctx, args.workload, args.steps, args.warmup, args.streams, args.seconds,
                                     outputs=args.outputs)
        if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ctx, state, args.cpu_streams)

    also_mode = args.also
    if also_mode == "auto":
        if args.workload != "spectrum_hop512" or args.streams or args.seconds:
            also_mode = "none"
        else:
            also_mode = "all" if ctx.world == 1 else "config4"
    if also_mode != "none":
        state = None
        torch.cuda.synchronize()
        import gc
        gc.collect()
        torch.cuda.empty_cache()                   # the chunked pass needs 217 GB of the 288
        also = []
        if also_mode == "all":
            for w in ("spectrum_1024", "spectrum_529", "spectrum_441", "lpf", "lpf_441", "biquad", "onset_2048",
                      "onset_2048_power"):
                r, st_ = run_resident(ctx, w, 3, 1)
                del st_
                gc.collect()
                torch.cuda.empty_cache()
                e = {"workload": w, "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                     "steps": 3, "description": r["config"]["workload"], "events_found": r["config"]["beats_found"],
                     "outputs": r["config"]["outputs"], "roofline": r["roofline"]}
                for k in ("value_decisions_only", "ms_per_step_decisions_only", "roofline_decisions_only"):
                    if k in r:
                        e[k] = r[k]
                also.append(e)
        also.append(run_chunked_onset(ctx, 2, 1, tracks=args.also_tracks, chunk=args.chunk_tracks))
        if also_mode == "all" and ctx.rank == 0:
            also.append(run_live(ctx))
        result["also"] = also
    if ctx.rank == 0:
        print(json.dumps(result), flush=True)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
