#!/usr/bin/env python
"""bench.py -- whole-job throughput of the hot path (rtl.c down-converter + msk.c demodulator +
framing FSM) on N GPUs of one node, with the roofline of the down-converter kernel and a CPU baseline.

A "step" is one pass of the hot path over one batch resident in HBM: `channels` independent u8 I/Q
streams per GPU (one stream per channel), `blocks` reference callbacks (1024 outputs each = 81.92 ms
of signal) per channel; every step ends with the decoded blocks delivered to the host.  The batch is
sized so that 20 steps make a timed region of >= 0.5 s.

`value` is BASELINE.json configs[2] (1024 channels x 2.5 Msps).  The same invocation also times, under
"also", the north-star regime (>= 10 000 independent channels on one GPU), BASELINE configs[4]
(4096 channels, 192-tap low-pass), configs[3]'s per-GPU share (2048 channels) and the other front ends'
sample formats, each with its own parity gate, down-converter roofline and whole-job fraction of HBM
bandwidth.

What is timed is what the product DELIVERS: contexts are created with ACG_F_REPAIR and every step ends with
acg_collect_msgs -- the block thread's check / repair (acars.c:93-215) and outputmsg()'s field split
(output.c:486-560) run on the device inside the timed region, and the gate compares the acg_msg records with
the oracle's orc_blk_process + orc_msg_split (--raw-blocks times the pre-repair blocks of rounds 1-3).

Output: the LAST stdout line is the compact result (< 4 KB: the driver keeps a bounded tail of stdout);
everything else -- per-case configuration, sustain / burst timing, telemetry, the full parity blocks -- is
printed on an EARLIER line prefixed "# bench_detail: " and written to bench_detail.json.

Multi-GPU: `python bench.py --gpus N` launches N ranks itself (torch.distributed.run, one rank per GPU,
backend nccl = RCCL); started under torchrun it uses the ranks it is given.  Channels are independent
(SURVEY 8e): rank r owns channels c = r (mod N), generates its input locally and keeps its own state;
there is no data-path collective.  RCCL carries the broadcast of the per-channel configuration table (32 B per channel; every rank keeps its rows), the
barriers around the timed region and the reductions of time and counts (scaling: weak).
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

# This process creates one context after the other on one device (a case each, plus the gate's exact-order context beside them).
# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues per device and priority, round robin, and
# streams that share a queue serialise against each other: which case's context drew aliasing streams followed the order of the
# cases (the same case: 2.45 M channel*Msps as the first of a run, 2.13 M as the fourth; profiles/LEDGER.md round 4).  So this
# host does what INTEGRATION.md tells every host with several contexts per device to do, before the runtime starts.  A process
# with ONE context is indifferent to the setting (profiles/r04_single_context_hw_queues.txt).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)
COPY_CEILING_GBS = 6290.0


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline_child(variant, M, blocks_per_call, seconds):
    """Times the UNMODIFIED reference's in_callback (rtl.c:314-361 incl. demodMSK/decodeAcars),
    one channel per stream, on one host core.  Runs in a child process: the reference is all
    global state, and an -march=native build may not run on this host."""
    import numpy as np
    from oracle import oracle as O
    from acarsdec_amd import synth as S
    ref = O.Ref(variant)
    ref.init_rtl(["131.725"], M)
    rng = np.random.default_rng(1)
    a, _ = S.channel_audio(rng, blocks_per_call * 1024)
    iq = S.iq_u8_from_envelopes(0.5 * (1 + 0.5 * a)[None, :], M, [-25000.0], noise=0.01, rng=rng)
    blk = 1024 * M * 2
    bufs = [np.ascontiguousarray(iq[b * blk:(b + 1) * blk]) for b in range(blocks_per_call)]
    for b in bufs:                      # warm-up
        ref.in_callback(b)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for b in bufs:
            ref.in_callback(b)
        n += len(bufs)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(value=n * 1024 * M / dt / 1e6, blocks=n, seconds=dt)))


def run_cpu_baseline(M, seconds=12.0):
    me = os.path.abspath(__file__)
    for variant, label in (("_fast", "-Ofast -march=native"), ("_v3", "-Ofast -march=x86-64-v3"), ("", "-O2")):
        so = os.path.join(ROOT, "oracle", "_ref", "libacarsref%s.so" % variant)
        if not os.path.exists(so):
            continue
        # 96 distinct callbacks = 39 MB per pass at rtlMult 200: the input streams from memory, as it does from a dongle (round 2
        # cycled through 4 cache-resident buffers, which flattered the CPU)
        cmd = [sys.executable, me, "--cpu-child", variant, str(M), "96", str(seconds)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0 and r.stdout.strip():
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out = dict(value=round(d["value"], 2), unit="channel*Msamples/s", cores=1, kind="reference",
                       sample="unmodified reference rtl.c in_callback + msk.c + acars.c (%s), 1 channel per stream, "
                              "rtlMult=%d, %d callbacks of 1024 outputs in %.1f s on one host core (the reference is "
                              "single-threaded; cycling through 96 distinct 410 KB callbacks = 39 MB, beyond the per-core caches)" % (label, M, d["blocks"], d["seconds"]))
            # the fair "all host cores" number: one independent reference process per core
            ncpu = min(os.cpu_count() or 1, 64)
            if ncpu > 1:
                cmd[-1] = str(max(4.0, seconds / 2))
                ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(ncpu)]
                tot = 0.0
                for p in ps:
                    o, _ = p.communicate()
                    if p.returncode == 0 and o.strip():
                        tot += json.loads(o.strip().splitlines()[-1])["value"]
                out["all_cores"] = dict(value=round(tot, 1), processes=ncpu)
            return out
    # no reference build travelled: time the C restatement instead
    import numpy as np
    from oracle import oracle as O
    iq = np.random.default_rng(0).integers(0, 256, size=1024 * M * 2, dtype=np.uint8)
    taps = O.rtl_taps(131725000, 131750000, M)
    ch = O.Channel(0)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ch.demod(O.fir_u8(iq, M, taps))
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=round(n * 1024 * M / dt / 1e6, 2), unit="channel*Msamples/s", cores=1, kind="port",
                sample="oracle/acars_oracle.c (-O2 IEEE), 1 channel, %d callbacks in %.1f s" % (n, dt))


# ------------------------------------------------------------------------------------------ workloads
# Every case is sized to 54-74 GB of input per GPU (one buffer serves them all); 20 steps of the headline case make >= 0.5 s
# (176 callbacks = 14.4 s of signal per channel per step).
CASES = {
    # BASELINE.json configs[2]: 1 GPU, 1024 channels, synthetic 2.5 Msps IQ, FIR decimate + MSK demod throughput
    "throughput": dict(tag="BASELINE configs[2]", channels=1024, decim=200, ntaps=200, blocks=176, content="acars"),
    # north-star regime: >= 10 000 concurrent channels at 2.5 Msps on one GPU
    "wide": dict(tag="north star (>= 10 000 channels per GPU)", channels=16384, decim=200, ntaps=200, blocks=8, content="acars"),
    # BASELINE.json configs[4]: 1 GPU stress, 192-tap LPF FIR, 2.5 Msps, 4096 channels
    "stress": dict(tag="BASELINE configs[4]", channels=4096, decim=200, ntaps=192, blocks=32, content="random+acars"),
    # the soapy.c front end's sample format (interleaved int16 I/Q) through the same pipeline
    "cs16": dict(tag="soapy.c CS16 front end (SURVEY 8f.2)", channels=4096, decim=200, ntaps=200, blocks=16, content="format+acars", format="cs16"),
    # ... and the air.c front end's (real float32 samples against complex taps)
    "f32": dict(tag="air.c real-f32 front end (SURVEY 8f.2)", channels=4096, decim=200, ntaps=200, blocks=16, content="format+acars", format="f32"),
    # BASELINE.json configs[3] per-GPU share: 16384 channels over 8 GPUs
    "shard2048": dict(tag="BASELINE configs[3], per-GPU share", channels=2048, decim=200, ntaps=200, blocks=64, content="acars"),
}
# The reference's own interface: the input arrives in HOST memory that is only valid during the call (rtl.c:314-330,
# soapy.c:220-254).  10 000 channels x 2.5 Msps is 50 GB/s -- PCIe Gen5 territory: this case says whether the ">= 10 000
# concurrent channels" of the north star holds for inputs that come from a host (run_hostfed below; not a run_case shape).
HOSTFED = dict(tag="north star fed from pinned host memory (rtl.c:314-330 semantics)", channels=10000, decim=200, ntaps=200, call_blocks=2)
SNR_DB = 20.0               # AWGN at 20 dB, measured in the 12.5 kHz channel
CARRIER, DEPTH, SCALE = 0.5, 0.5, 0.25


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks here."""
    import torch
    backend = os.environ.get("ACG_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < args.gpus and backend != "gloo":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (ACG_BENCH_BACKEND=gloo rehearses the launch path "
                         "with several ranks per GPU)" % (args.gpus, ndev))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


_CARD_DIR = {}


def _card_dir(local):
    """sysfs directory of HIP device `local`: by PCI address where torch exposes it, else the first amdgpu card"""
    if local in _CARD_DIR:
        return _CARD_DIR[local]
    import glob
    d = None
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        if os.path.exists("/sys/bus/pci/devices/%s/pp_dpm_sclk" % bdf):
            d = "/sys/bus/pci/devices/%s" % bdf
    except Exception:
        d = None
    if d is None:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        d = os.path.dirname(cards[min(local, len(cards) - 1)]) if cards else None
    _CARD_DIR[local] = d
    return d


def _starred(path):
    try:
        with open(path) as f:
            for line in f:
                if "*" in line:
                    return int(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
    except Exception:
        return None
    return None


def gpu_clock_mhz(local):
    """current shader clock of GPU `local` from sysfs (amdgpu pp_dpm_sclk: the starred level), or None"""
    d = _card_dir(local)
    return _starred(os.path.join(d, "pp_dpm_sclk")) if d else None


def gpu_telemetry(local):
    """{sclk, fclk, mclk (MHz), power (W)} of GPU `local` from sysfs; what cannot be read is None"""
    import glob
    d = _card_dir(local)
    if not d:
        return dict(sclk=None, fclk=None, mclk=None, power_w=None)
    pw = None
    for h in glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input")):
        try:
            pw = round(int(open(h).read().strip()) / 1e6, 1)
            break
        except Exception:
            pass
    return dict(sclk=_starred(os.path.join(d, "pp_dpm_sclk")), fclk=_starred(os.path.join(d, "pp_dpm_fclk")),
                mclk=_starred(os.path.join(d, "pp_dpm_mclk")), power_w=pw)


class Job:
    """Process-wide state: device, library, collectives, the one input buffer."""
    pass


def make_taps(D, fmt_name, offs, M, ntaps):
    import numpy as np
    fc = 131000000
    nch = len(offs)
    taps = np.zeros((nch, ntaps, 2), dtype=np.float32)
    # ntaps < M: a low-pass window over the NCO (Hamming, unit DC gain); the oracle for it is the same
    # sum(vb*wf) with these taps (SURVEY 8d config 5 -- the reference itself only has the boxcar)
    win = np.ones(ntaps) if ntaps == M else np.hamming(ntaps) / np.hamming(ntaps).mean() * (M / ntaps)
    cache = {}
    for c in range(nch):
        o = int(offs[c])
        if o not in cache:
            # the front end's own tap builder (rtl.c:283-286 / soapy.c:163-166 / air.c:278-285)
            base = (D.rtl_taps(fc + o, fc, M) if fmt_name == "u8" else
                    D.airspy_taps(fc - o, fc, M * 12500) if fmt_name == "f32" else D.soapy_taps(fc + o, fc, M))
            cache[o] = (base[:ntaps] * win[:, None]).astype(np.float32)
        taps[c] = cache[o]
    return taps


def lookup_traffic(kernel, nch, M, ntaps, blocks_per_launch):
    """HBM bytes of one launch of this shape from the committed PMC passes (profiles/pmc_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs of the same command; rocprofv3 cannot run inside the timed process): (bytes, source) or (None, None)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if (e["channels"], e["decim"], e["ntaps"]) == (nch, M, ntaps) and abs(e["blocks_per_launch"] - blocks_per_launch) < 1e-9 \
                        and e["kernel"] == kernel:
                    return e["traffic_bytes"], e.get("source", "profiles/pmc_traffic.json")
    except Exception:
        pass
    return None, None


def live_traffic(case_name, kernel, nch, blocks_per_launch, timeout_s=100):
    """HBM bytes of ONE launch of `kernel` measured in THIS invocation: two child runs of this script under rocprofv3
    (--kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE: separate passes, never combined with another trace domain, as
    MI355X_MICROARCH.md's HBM section prescribes), a short burst of the same launch shape each; traffic = 2 x FETCH_SIZE x 1024
    + WRITE_SIZE x 1024 (gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes; profiles/pmc_traffic.json _about).
    Returns (bytes, source text) or (None, reason): any failure leaves the committed look-up in place."""
    import shutil
    import sqlite3
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    # (this process may itself be running under a profiler -- `rocprofv3 --stats -- python bench.py ...`: no profiler inside a profiler)
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler: no nested PMC pass"
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                cmd = [prof, "--kernel-trace", "--pmc", ctr, "-d", td, "--", sys.executable, os.path.abspath(__file__), "--config", case_name,
                       "--also", "none", "--blocks", str(int(2 * blocks_per_launch)), "--channels", str(nch), "--steps", "3", "--warmup", "1", "--sustain", "0",
                       "--no-cpu-baseline", "--no-ref-leg", "--check-channels", "8", "--no-live-traffic", "--detail-file", os.path.join(td, "detail.json")]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(td) for f in fs if f.endswith(".db")]
                if r.returncode != 0 or not dbs:
                    return None, "rocprofv3 --pmc %s child failed (%d)" % (ctr, r.returncode)
                con = sqlite3.connect(dbs[0])
                rows = con.execute("select k.name, count(*), avg(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                                   "where p.counter_name = ? group by k.name", (ctr,)).fetchall()
                con.close()
                hit = [(n, c, v) for n, c, v in rows if n.replace("void ", "").startswith(kernel.split("(")[0])]
                if not hit:
                    return None, "kernel %s not in the %s pass" % (kernel, ctr)
                vals[ctr] = (hit[0][2], hit[0][1])
    except Exception as ex:                      # (timeouts included: the line must not depend on a profiler)
        return None, "live PMC pass failed: %r" % (ex,)
    traffic = int(round(2 * vals["FETCH_SIZE"][0] * 1024 + vals["WRITE_SIZE"][0] * 1024))
    return traffic, ("measured in this invocation: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child runs of this script at "
                     "the same launch shape, %d / %d launches): 2 x FETCH_SIZE + WRITE_SIZE" % (vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]))


def _probe_ab(args, J, step, drain, steps, nch, nout, M):
    """measurement aid (--ab): the same decoder, buffers and placement, timed again under each value of a per-launch switch in
    turn (acg_tune: ACG_FIR_VARIANT, ACG_MSK_LPC_LIVE, ...), two rounds -- not part of the reported value"""
    import torch
    from acarsdec_amd import _capi as K
    ab = {}
    ab_name, _, ab_vals = args.ab.rpartition("=")                # "5,55,8" or "ACG_MSK_LPC_LIVE=2,4"
    ab_name = ab_name or "ACG_FIR_VARIANT"
    for rnd in range(2):
        for v in ab_vals.split(","):
            K.tune(ab_name, v)
            step()
            drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            clk_ab = None
            for i_ in range(steps):
                step()
                if i_ == steps - 2:
                    clk_ab = gpu_telemetry(J.local)
            drain()
            torch.cuda.synchronize()
            ab.setdefault(v, []).append(round(nch * nout * M * steps / (time.perf_counter() - t1) / 1e6, 0))
            ab.setdefault(v + " telemetry", []).append(clk_ab)
    K.tune(ab_name, os.environ.get(ab_name))
    return ab


def _probe_decoders(args, J, make_decoder, step, drain, steps, reps, dt_local, nch, nout, M, dev):
    """measurement aid (--decoders N): further decoders in the same process (each with its own allocations, all kept alive), the
    same input, timed the same way -- how much of the run-to-run spread is where the decoder's buffers happen to lie.  Probe
    switches: ACG_BENCH_SPACER_MB changes where the next decoder's buffers land without touching its streams;
    ACG_BENCH_DUMMY_STREAMS creates streams in between, which shifts the decoder's streams to other hardware queues without
    touching its memory; ACG_BENCH_DECODERS_ALT times each decoder once more under another FIR variant"""
    import torch
    from acarsdec_amd import _capi as K
    trials = [round(nch * nout * M * steps * reps / dt_local / 1e6, 0)]
    others, spacers, dummies, trials_alt = [], [], [], []
    for k in range(1, args.decoders):
        sp = int(os.environ.get("ACG_BENCH_SPACER_MB", "53"))
        if sp:
            spacers.append(torch.empty((((k * sp) << 20) + 4096 * k,), dtype=torch.uint8, device=dev))
        for _ in range(int(os.environ.get("ACG_BENCH_DUMMY_STREAMS", "0"))):
            dummies.append(torch.cuda.Stream(priority=-1))
            dummies.append(torch.cuda.Stream())
        d2 = make_decoder()
        others.append(d2)

        def timed():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                step(dec=d2)
            drain(d2)
            torch.cuda.synchronize()
            return round(nch * nout * M * steps / (time.perf_counter() - t1) / 1e6, 0)
        for _ in range(2):
            step(dec=d2)
        drain(d2)
        trials.append(timed())
        alt = os.environ.get("ACG_BENCH_DECODERS_ALT")
        if alt:
            K.tune("ACG_FIR_VARIANT", alt)
            step(dec=d2)
            drain(d2)
            trials_alt.append(timed())
            K.tune("ACG_FIR_VARIANT", os.environ.get("ACG_FIR_VARIANT"))
    for d2 in others:
        d2.close()
    return dict(default=trials, alt_variant=trials_alt) if trials_alt else trials


def run_case(J, name, case, args, steps, warmup, headline):
    """Builds the input of one workload in J.iq, checks the first pass against the oracle, times `steps`
    steps.  Returns the dict that goes into the JSON line (rank 0) or None."""
    import numpy as np
    import torch
    from acarsdec_amd import decoder as D, synth as S, _capi as K, shard
    L, dist, world, rank, dev, cdev = J.L, J.dist, J.world, J.rank, J.dev, J.cdev
    nch, M, ntaps, nblk, content = case["channels"], case["decim"], case["ntaps"], case["blocks"], case["content"]
    fmt_name = case.get("format") or (args.format if headline else "u8")
    fmt = {"u8": 0, "cs16": K.FMT_CS16, "split16": K.FMT_S16_SPLIT, "f32": K.FMT_F32_REAL}[fmt_name]
    bps = 2 if fmt == 0 else 4
    share = max(1, args.share) if headline else 1
    if share > 1:
        assert fmt == 0 and nch % share == 0, "--share needs the u8 format and a channel count divisible by it"
        content = "random"
    if fmt != 0 and content != "format+acars":
        content = "format"
    nstreams = nch // share
    nout = nblk * 1024
    row = nout * M * bps
    assert nstreams * row <= J.iq_all.numel(), "input buffer too small for this case"
    iq = J.iq_all[: nstreams * row].view(nstreams, row)

    # ---- per-channel configuration: made on rank 0 for ALL channels of the job, scattered over RCCL
    # (the only data that ever crosses xGMI: 32 B per channel; inputs are generated where they are used)
    nch_total = nch * world
    cfg_rows = None
    if rank == 0:
        r0 = np.random.default_rng(0xACA25)
        off = r0.integers(-48, 49, size=nch_total) * 25000.0           # multiples of 12.5 kHz within +-1.2 MHz
        off[np.abs(off) < 25000] = 50000.0                              # >= 25 kHz from DC like chooseFc enforces
        cfg_rows = np.stack([off, r0.uniform(0, 2 * np.pi, nch_total), np.zeros(nch_total),
                             np.arange(nch_total, dtype=np.float64)], axis=1)
    mine = shard.scatter_channel_config(cfg_rows, world, rank, J.coll, device=cdev, force=J.coll is not None)
    own = shard.owned_channels(nch_total, rank, world)
    assert mine.shape[0] == nch and np.array_equal(mine[:, 3].astype(np.int64), own)
    offs, phases = mine[:, 0], mine[:, 1]
    taps = make_taps(D, fmt_name, offs, M, ntaps)

    # ---- input, resident in HBM: distinct content per channel, working set >> 256 MiB Infinity Cache
    sigma = SCALE * CARRIER * (M / (2.0 * 10 ** (SNR_DB / 10.0))) ** 0.5

    def synth_acars(n_first):
        """channels [0, n_first) of this rank: ACARS/MSK traffic, seeded 0xACA25 + global channel
        id: random printable frames of 20-220 characters every 0.25-1 s, AM depth 0.5, own carrier offset and phase, AWGN at
        20 dB SNR in the 12.5 kHz channel; modulated on the host (numpy), up-converted and quantised on the device."""
        trk = torch.empty((n_first, nout), dtype=torch.float32, device=dev)
        GEN = 512
        for c0 in range(0, n_first, GEN):
            n = min(GEN, n_first - c0)
            buf = np.empty((n, nout), dtype=np.float32)
            for i in range(n):
                a, _ = S.channel_audio(np.random.default_rng(0xACA25 + int(own[c0 + i])), nout, gap=(3125, 12500), text_len=(20, 220))
                buf[i] = CARRIER * (1.0 + DEPTH * a)
            trk[c0:c0 + n] = torch.from_numpy(buf).to(dev)
        d_idx = torch.arange(n_first, dtype=torch.int32, device=dev)
        d_off = torch.from_numpy(offs[:n_first].astype(np.float32)).to(dev)
        d_ph = torch.from_numpy(phases[:n_first].astype(np.float32)).to(dev)
        rc = L.acg_synth_iq_u8_dev(iq.data_ptr(), row, n_first, nout, M, trk.data_ptr(), nout, d_idx.data_ptr(),
                                   d_off.data_ptr(), d_ph.data_ptr(), SCALE, sigma, 0xACA25 + rank, None)
        assert rc == 0, rc
        torch.cuda.synchronize()

    if content == "acars":
        synth_acars(nch)
        data_desc = ("ACARS/MSK traffic on every channel, content seeded 0xACA25 + channel id (frames of 20-220 characters every "
                     "0.25-1 s), AM depth %.1f, carrier offset and phase per channel, AWGN at %.0f dB SNR in the 12.5 kHz channel "
                     "(sigma %.4f per I/Q sample); MSK modulator on the host, up-converter + u8 quantiser on the device" % (DEPTH, SNR_DB, sigma))
    elif content == "random+acars":
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
        nacars = min(nch, max(64, args.check_channels))
        synth_acars(nacars)
        data_desc = ("uniform random bytes, seeded per stream (SURVEY 8d config 5: the value distribution is irrelevant to bandwidth); "
                     "the first %d channels (the ones the parity gate looks at) carry ACARS/MSK traffic as in the other cases, so "
                     "that the gate compares decoded blocks and not only magnitudes" % nacars)
    elif content == "format+acars":
        assert fmt in (K.FMT_CS16, K.FMT_F32_REAL)
        if fmt == K.FMT_CS16:
            assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
            iq.view(torch.int16).bitwise_and_(0x0FFF)
        else:
            iq.view(torch.float32).normal_(0.0, 0.1)
        nacars = min(nch, max(64, args.check_channels))
        # the gate's channels: ACARS/MSK traffic as in the u8 cases, up-converted with torch on the device, a few channels at a time:
        # CS16 = complex baseband quantised to int16 (rint(32767 * 0.9 x), synth.iq_s16_from_envelopes); real f32 = 2 x env x cos
        # at the channel's offset from 0 Hz of the real spectrum (synth.real_f32_from_envelopes)
        tt = torch.arange(nout * M, dtype=torch.float64, device=dev) * (2.0 * np.pi / (12500.0 * M))
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xACA25 + rank)
        v16 = iq.view(torch.int16).view(nstreams, -1)
        v32 = iq.view(torch.float32).view(nstreams, -1)
        for c in range(nacars):
            a, _ = S.channel_audio(np.random.default_rng(0xACA25 + int(own[c])), nout, gap=(3125, 12500), text_len=(20, 220))
            env = torch.from_numpy((SCALE * CARRIER * (1.0 + DEPTH * a)).astype(np.float32)).to(dev).repeat_interleave(M)
            # (real f32: air.c mixes with Fc - Fr + rate / 4, air.c:278, i.e. the channel sits at its offset + a quarter of the rate)
            f_c = float(offs[c]) + (12500.0 * M / 4.0 if fmt == K.FMT_F32_REAL else 0.0)
            ph = torch.remainder(tt * f_c + float(phases[c]), 2.0 * np.pi).to(torch.float32)
            if fmt == K.FMT_CS16:
                xi = env * torch.cos(ph) + sigma * torch.randn(nout * M, device=dev, generator=gen)
                xq = env * torch.sin(ph) + sigma * torch.randn(nout * M, device=dev, generator=gen)
                v16[c, 0::2] = torch.round(32767.0 * 0.9 * xi).clamp_(-32768, 32767).to(torch.int16)
                v16[c, 1::2] = torch.round(32767.0 * 0.9 * xq).clamp_(-32768, 32767).to(torch.int16)
            else:
                v32[c] = 2.0 * env * torch.cos(ph) + sigma * torch.randn(nout * M, device=dev, generator=gen)
        del tt, env, ph
        data_desc = ("%s; the first %d channels (the ones the parity gate looks at) carry ACARS/MSK traffic "
                     "as in the u8 cases (AM depth %.1f, %.0f dB SNR in the channel), generated on the device"
                     % ("uniform random 12-bit int16 samples" if fmt == K.FMT_CS16 else "gaussian float32 samples", nacars, DEPTH, SNR_DB))
    elif content == "random":
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
        data_desc = "uniform random bytes, seeded per stream (SURVEY 8d config 5: the value distribution is irrelevant to bandwidth)"
    else:
        if fmt == K.FMT_F32_REAL:
            iq.view(torch.float32).normal_(0.0, 0.1)
            data_desc = "gaussian float32 samples (format throughput run; blocks of this format are covered by tests/)"
        else:
            assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
            iq.view(torch.int16).bitwise_and_(0x0FFF)
            data_desc = "uniform random 12-bit int16 samples (format throughput run; blocks of this format are covered by tests/)"
    torch.cuda.synchronize()

    # The batch is streamed through the library in calls of `cb` callbacks (the reference hands over ONE callback at a
    # time, rtl.c:314; 8 keeps the 12.5 kHz intermediate of a call inside the Infinity Cache at 1024 channels).
    cb = min(args.call_blocks, nblk)
    while nblk % cb:
        cb -= 1
    ncall = nblk // cb
    if fmt == K.FMT_S16_SPLIT:
        cb, ncall = nblk, 1                                # (plane layout: one call)
    repair = not args.raw_blocks
    def make_decoder():
        # max_lag = --collect-lag: this host collects that many calls behind, and the block queue holds that many + 1 calls' worth
        d_ = D.Decoder(nch, decim=M, ntaps=ntaps, nstreams=nstreams, max_blocks=cb, device=J.local, bitlog=bool(args.bitlog), timing=True,
                       repair=repair, max_lag=max(1, args.collect_lag))
        d_.set_taps(taps)
        if share > 1:
            d_.set_channel_streams(np.arange(nch) // share)
        return d_
    stream = torch.cuda.current_stream().cuda_stream
    # The context a host gets from acg_create is the one that is timed (--placements 1, the default).  --placements N is a
    # DIAGNOSTIC: N contexts alive at once, acg_placement_trial on each, their times reported under config.placement; the
    # FIRST is still the one timed unless --placement-keep best (rounds 2-3 kept the fastest of four: selection).
    ntrial = args.placements if share == 1 else 1
    dec, trial_ms, trial_best = D.best_placed(make_decoder, ntrial, iq, cb, row, repeats=8 if nch > 2048 else 24, stream=stream, fmt=fmt,
                                              plane=row // 2 if fmt == K.FMT_S16_SPLIT else 0, keep=args.placement_keep)
    dec0 = dec
    maxfr = max(8192, int(nch * (cb / 3.0 + 2)))
    cb_bytes = cb * 1024 * M * bps

    def step(lag=None, sink=None, dec=None, dm_sink=None, frames=False):
        """one pass of the hot path over the batch; the results (acg_msg records; blocks with frames=True or --raw-blocks) are
        delivered to the host --collect-lag calls behind (default 2: the host then never waits for the block repair of the call
        before the newest one before it may hand over the next -- with a lag of 1 that wait sat between every two calls)"""
        lag = args.collect_lag if lag is None else lag
        n = 0
        dec = dec or dec0
        for k in range(ncall):
            part = iq[:, k * cb_bytes:(k + 1) * cb_bytes]
            if fmt == 0:
                dec.in_callback(part, nblocks=cb, pitch=row, stream=stream)
            else:
                dec.process_samples(fmt, part, cb, pitch=row, plane=row // 2 if fmt == K.FMT_S16_SPLIT else 0, stream=stream)
            if repair and not frames:
                # the delivered path: repaired blocks through outputmsg()'s field split, as acg_msg records
                m = 0
                while True:
                    mm, fb, more = dec.collect_msgs_raw(lag, maxfr)
                    if sink is not None:
                        sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(mm)]
                    m += mm
                    if not more:
                        break
            else:
                m, fb = dec.collect_frames_raw(lag, maxfr)
                if sink is not None:
                    sink += [K.Frame.from_buffer_copy(fb[i]) for i in range(m)]
            if dm_sink is not None:                     # (gate only) the 12.5 kHz samples this call's demodulator consumed
                for c in dm_sink:
                    dm_sink[c].append(dec.dm(c, cb * 1024))
            n += m
        return n

    def drain(dec=None):
        """everything still queued, through the delivered path; returns the count"""
        dec = dec or dec0
        if not repair:
            return dec.drain_frames_raw(maxfr)[0]
        m = 0
        while True:
            mm, _, more = dec.drain_msgs_raw(maxfr)
            m += mm
            if not more:
                return m

    def barrier():
        torch.cuda.synchronize()
        if J.coll is not None:
            dist.barrier(device_ids=[J.local]) if J.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # ---- correctness gate on the first pass (state starts from reset): a subset of this rank's channels goes through
    # the CPU checkers on the very bytes the GPU consumed.  SURVEY 8c's parity statement has two halves, and the gate
    # checks each of them and then closes the argument between them:
    #   (1) the 12.5 kHz magnitudes of EVERY call against the oracle's down-converter: |d dm| <= 1e-5 |dm| + 1e-6 full scale
    #       (the streaming kernel re-associates the sum; so does the reference's own -Ofast build);
    #   (2) the blocks against the oracle's demodulator + framing fed with the dm the GPU's demodulator consumed: BIT-EXACT
    #       (`blocks_exact_given_gpu_dm`: the demodulator and the framing are exact);
    #   (3) the same channels once more through the library in its exact-order mode (ACG_F_EXACT_FIR: rtl.c:335-353 in the
    #       reference's own order of operations): dm BIT-IDENTICAL to the oracle's, blocks identical END TO END -- so the only
    #       thing that can differ between the product path and the reference is the rounding of (1);
    #   (4) end to end with the streaming kernel (oracle down-converter -> oracle demodulator): a 1e-7 difference in dm can
    #       flip a soft decision that sits at |vo| < 1e-3 in a noise-only stretch, after which the two loops wander apart until
    #       the next preamble and one of them may lock a block late.  How often the reference's own builds do that to each
    #       other is MEASURED here: the same bytes and taps through the unmodified reference compiled -O2 (IEEE) and with its
    #       own flags (-Ofast -march=native), both from oracle/_ref.  The streaming path may differ from the oracle in no more
    #       blocks than those two builds differ from each other, plus one.
    #   (5) the DELIVERED records: the pass once more from reset, collected as acg_msg (ACG_F_REPAIR + acg_collect_msgs), against
    #       orc_blk_process + orc_msg_split of the oracle's blocks of (2): every field of every message, and no message of a
    #       block that the reference's block thread drops (acars.c:124-207).
    # With ACG_F_REPAIR (the default) "blocks" are what outputmsg() receives: checked / repaired, parity stripped, the dropped
    # ones omitted -- on both sides (oracle: orc_blk_process; reference builds: what their blk_thread handed to outputmsg()).
    def gate_first_pass():
        """the first pass from reset through the CPU checkers (the comment above); returns the parity record (rank 0) or None;
        raises SystemExit when the GPU output differs.  Nothing in here is timed."""
        parity = None
        first = []
        ncheck = min(args.check_channels, nch) if rank == 0 else 0
        dm_gpu = {c: [] for c in range(ncheck)}
        step(lag=0, sink=first, dm_sink=dm_gpu if ncheck else None, frames=True)
        msgs_first = []
        if repair:
            dec.reset()
            step(lag=0, sink=msgs_first)
        parity = None
        if rank == 0:
            from oracle import oracle as O

            def processed(frames):
                """the oracle's block thread on raw blocks: kept ones as OrcFrame (ACG_F_REPAIR), or the raw blocks themselves"""
                if not repair:
                    return list(frames)
                return [b for b in (O.blk_process(f) for f in frames) if b is not None]
            got = {}
            got_end = {}
            for f in first:
                got.setdefault(int(f.chn), []).append(D.frame_tuple(f))
                got_end.setdefault(int(f.chn), []).append(int(f.end_bit))
            got_msgs = {}
            for m_ in msgs_first:
                got_msgs.setdefault(int(m_.chn), []).append(O.msg_tuple(m_))
            ok, nblocks, dm_err, dm_ok = True, 0, 0.0, True
            msgs_ok, nmsgs, nraw, first_bad_msg = True, 0, 0, None
            e2e_blocks_off, e2e_channels_off = 0, []
            first_bad = None
            # absolute floor of the dm tolerance: 1e-6 of the largest term of the sum.  u8: |x - 127.37| / 127.5 <= 1; CS16:
            # 4095 / 32768; split planes (random 12-bit samples, |D| / 4): 4095 / 4; real f32: ~0.5
            dm_fullscale = {0: 1.0, K.FMT_CS16: 1.0, K.FMT_S16_SPLIT: 1024.0, K.FMT_F32_REAL: 1.0}[fmt]
            host_rows = iq[:(ncheck + share - 1) // share].cpu().numpy()
            dm_orc, e2e_want = [], []
            for c in range(ncheck):
                r = host_rows[c // share]
                if fmt == 0:
                    dm = O.fir_u8(r, M, taps[c], ntaps=ntaps)
                elif fmt == K.FMT_CS16:
                    dm = O.fir_cs16(r.view(np.int16), M, taps[c])
                elif fmt == K.FMT_S16_SPLIT:
                    h = r.view(np.int16)
                    dm = O.fir_split16(h[: h.size // 2], h[h.size // 2:], M, taps[c])
                else:
                    dm = O.fir_f32r(r.view(np.float32), M, taps[c])
                dm_orc.append(dm)
                g = np.concatenate(dm_gpu[c])
                e = np.abs(g - dm[: g.size])
                dm_ok &= bool(g.size == dm.size and np.all(e <= 1e-5 * np.abs(dm) + 1e-6 * dm_fullscale))
                dm_err = max(dm_err, float(e.max()))
                ch = O.Channel(c)
                ch.demod(g)                                         # (2): the oracle's demodulator on the GPU's dm
                nraw += len(ch.frames)
                kept = processed(ch.frames)
                want = [O.frame_tuple(f) for f in kept]
                nblocks += len(want)
                mine = got.get(c, [])
                if mine != want and first_bad is None:
                    k_ = next((i for i in range(min(len(mine), len(want))) if mine[i] != want[i]), min(len(mine), len(want)))
                    first_bad = dict(channel=c, gpu_blocks=len(mine), oracle_blocks=len(want), first_difference_at=k_,
                                     gpu_end_bits=got_end.get(c, []), oracle_end_bits=[int(f.end_bit) for f in ch.frames],
                                     gpu=repr(mine[k_])[:300] if k_ < len(mine) else None, oracle=repr(want[k_])[:300] if k_ < len(want) else None)
                ok &= mine == want
                if repair:                                          # (5): the delivered records, field for field
                    want_m = [O.msg_tuple(O.msg_split(b)) for b in kept]
                    nmsgs += len(want_m)
                    mine_m = got_msgs.get(c, [])
                    if mine_m != want_m and first_bad_msg is None:
                        first_bad_msg = dict(channel=c, gpu_msgs=len(mine_m), oracle_msgs=len(want_m))
                    msgs_ok &= mine_m == want_m
                ch2 = O.Channel(c)
                ch2.demod(dm)                                       # (4): oracle down-converter -> oracle demodulator
                want2 = [O.frame_tuple(f) for f in processed(ch2.frames)]
                e2e_want.append(want2)
                if mine != want2:
                    e2e_channels_off.append(c)
                    e2e_blocks_off += len(set(mine) ^ set(want2))
            # (3) the exact-order mode of the library on the same channels
            exact = None
            if fmt == 0 and share == 1 and ncheck:
                dx = D.Decoder(ncheck, decim=M, ntaps=ntaps, nstreams=ncheck, max_blocks=cb, device=J.local, bitlog=False, exact_fir=True, repair=repair)
                dx.set_taps(taps[:ncheck])
                xfr, xdm_same = [], True
                for k in range(ncall):
                    dx.in_callback(iq[:ncheck, k * cb_bytes:(k + 1) * cb_bytes], nblocks=cb, pitch=row, stream=stream)
                    for c in range(ncheck):
                        xdm_same &= bool(np.array_equal(dx.dm(c, cb * 1024).view(np.uint32), dm_orc[c][k * cb * 1024:(k + 1) * cb * 1024].view(np.uint32)))
                xgot = {}
                for f in dx.drain_frames(maxfr):
                    xgot.setdefault(int(f.chn), []).append(D.frame_tuple(f))
                dx.close()
                xoff = sum(len(set(xgot.get(c, [])) ^ set(e2e_want[c])) for c in range(ncheck))
                xsame = all(xgot.get(c, []) == e2e_want[c] for c in range(ncheck))
                exact = dict(dm_bit_identical_to_oracle=bool(xdm_same), blocks=sum(len(w) for w in e2e_want),
                             blocks_differing_end_to_end=int(xoff), blocks_identical_end_to_end=bool(xsame),
                             means="the library in ACG_F_EXACT_FIR mode (rtl.c:335-353 in the reference's order) -> the same GPU demodulator: "
                                   "everything identical to oracle down-converter -> oracle demodulator, so the streaming path's only deviation is "
                                   "the re-associated sum of its down-converter")
            # (4b) the reference's own builds against each other on the same bytes and taps: rtl.c in_callback for u8, soapy.c's reader
            # loop for CS16, air.c rx_callback for real f32 (oracle/_ref: the unmodified sources, -O2 and the reference's -Ofast)
            refs = None
            front = {0: "rtl", K.FMT_CS16: "soapy", K.FMT_F32_REAL: "air"}.get(fmt)
            if front and share == 1 and ncheck and not args.no_ref_leg:
                rows_ = [host_rows[c] for c in range(ncheck)]
                wf_ = [taps[c] for c in range(ncheck)]
                t_ref = time.perf_counter()
                which = "out" if repair else "raw"
                pick = lambda d: None if d is None else d[which]
                if front == "rtl":
                    b_o2 = pick(O.ref_blocks("", rows_, M, wf_))
                    b_fast, fast_label = pick(O.ref_blocks("_fast", rows_, M, wf_)), "-Ofast -march=native"
                    if b_fast is None:
                        b_fast, fast_label = pick(O.ref_blocks("_v3", rows_, M, wf_)), "-Ofast -march=x86-64-v3"
                else:
                    b_o2 = pick(O.ref_blocks("_" + front, rows_, M, wf_, front=front))
                    b_fast, fast_label = pick(O.ref_blocks("_%s_fast" % front, rows_, M, wf_, front=front)), "-Ofast -march=x86-64-v3"
                if b_o2 is not None and b_fast is not None:
                    strip = lambda lst: [t[1:] for t in lst]
                    refs = dict(o2_blocks=sum(len(x) for x in b_o2), ofast_blocks=sum(len(x) for x in b_fast),
                                ref_fast_vs_ref_o2_blocks_differing=sum(len(set(x) ^ set(y)) for x, y in zip(b_o2, b_fast)),
                                oracle_vs_ref_o2_blocks_differing=sum(len(set(x) ^ set(strip(y))) for x, y in zip(b_o2, e2e_want)),
                                gpu_vs_ref_o2_blocks_differing=sum(len(set(x) ^ set(strip(got.get(c, [])))) for c, x in enumerate(b_o2)),
                                gpu_vs_ref_ofast_blocks_differing=sum(len(set(x) ^ set(strip(got.get(c, [])))) for c, x in enumerate(b_fast)),
                                builds="oracle/_ref (-O2, IEEE) vs the reference's own flags (%s): unmodified %s + msk.c + "
                                       "acars.c (%s) on the GPU's input bytes and tap tables, one channel per pass, each build in a child interpreter"
                                       % (fast_label, {"rtl": "rtl.c in_callback", "soapy": "soapy.c reader loop", "air": "air.c rx_callback"}[front],
                                          "blocks as its blk_thread hands them to outputmsg()" if repair else "blocks as decodeAcars queues them"),
                                cpu_seconds=round(time.perf_counter() - t_ref, 1))
            # What the streaming path may differ from the IEEE oracle by: exactly what the reference's own -O2 and -Ofast builds differ
            # from each other on these bytes (MEASURED above; no slack on top of it -- VERDICT r04), and, where the -Ofast leg ran, NOT
            # AT ALL from the reference as shipped (its -Ofast build).  Without a reference leg (split planes, shared streams,
            # --no-ref-leg, oracle/_ref absent) that yardstick is missing: (4) is then reported, not enforced -- (1)-(3) are, and they
            # already pin the only deviation of the streaming path to the rounding of (1).
            allowed = refs["ref_fast_vs_ref_o2_blocks_differing"] if refs else None
            parity = dict(channels_checked=ncheck, blocks=nblocks, blocks_exact_given_gpu_dm=bool(ok),
                          blocks_are=("what outputmsg() receives: checked / repaired by the device (ACG_F_REPAIR, acars.c:93-215), parity stripped, "
                                      "dropped blocks omitted" if repair else "as decodeAcars queues them (pre-repair, --raw-blocks)"),
                          raw_blocks_before_repair=nraw,
                          blocks_exact_given_gpu_dm_means="blocks identical to the oracle's demodulator + framing (+ block repair) fed with the dm the GPU's demodulator consumed",
                          msgs=(dict(records=nmsgs, exact=bool(msgs_ok), delivered=len(msgs_first),
                                     means="acg_msg records of acg_collect_msgs (a second pass from reset) == orc_msg_split(orc_blk_process(block)) field for field "
                                           "(output.c:486-560)") if repair else None),
                          dm_within_1e5_rel=bool(dm_ok), dm_max_abs_err=dm_err, dm_samples_per_channel=nout,
                          exact_order_mode=exact,
                          end_to_end=dict(blocks_differing=e2e_blocks_off, channels=e2e_channels_off, exact=bool(e2e_blocks_off == 0),
                                          allowed=allowed, allowed_means="what the reference's -O2 and -Ofast builds differ by on this input (measured in this run); "
                                                                         "and zero against the reference's -Ofast build",
                                          gpu_vs_ref_ofast=(refs["gpu_vs_ref_ofast_blocks_differing"] if refs else None),
                                          note="streaming down-converter -> GPU demodulator against oracle down-converter -> oracle demodulator; a differing "
                                               "block = a razor-edge soft decision (|vo| < 1e-3 in noise) flipped by the 1e-7 re-association of dm"),
                          reference_builds=refs,
                          blocks_first_pass_all_channels=len(first))
            bad = (not (ok and dm_ok and msgs_ok) or (allowed is not None and e2e_blocks_off > allowed) or
                   (refs is not None and (refs["gpu_vs_ref_ofast_blocks_differing"] != 0 or refs["oracle_vs_ref_o2_blocks_differing"] != 0)) or
                   (exact is not None and not (exact["dm_bit_identical_to_oracle"] and exact["blocks_identical_end_to_end"])))
            if bad:
                raise SystemExit("bench[%s]: GPU output differs from the oracle: %r; first mismatch: %r %r" % (name, parity, first_bad, first_bad_msg))
        return parity

    parity = gate_first_pass()

    # ---- timing.  A "pass" = the hot path once over the resident batch (the step of rounds 1-2).  `burst`: `steps` single
    # passes, timed as before (about half a second at the headline case: too short to be seen by an outside observer, and
    # inside the window in which the shader clock has not settled).  The reported `value` is SUSTAINED: a step is `reps`
    # passes, reps chosen from the burst rate so that `steps` steps take >= --sustain seconds; per-step times (host clock at
    # the step boundaries, no extra synchronisation: the host runs at most one call ahead of the device) give min / median /
    # max, the shader clock is read from sysfs while the device is still busy.
    def timed_region():
        """warm-up, the burst of `steps` single passes, then the reported region: `steps` steps of `reps` passes each, bracketed
        by barrier + synchronize on both sides; everything a step does is inside step() / drain() above: the process call(s) of
        the hot path and the collect of the delivered records.  Returns the raw clocks and counters; no probe, no switch."""
        for _ in range(warmup):
            step()
        drain()                           # flush: the timed region starts with empty queues
        warm = dec.timing()               # event sums of warm-up: the demodulator's launches are timed here only --
        dec.set_timing(2)                 # in the timed region only the down-converter (roofline) is bracketed,
                                          # event records on the demodulator stream sit on its serial launch chain
        barrier()
        t0 = time.perf_counter()
        nfr_b = 0
        for _ in range(steps):
            nfr_b += step()
        nfr_b += drain()
        barrier()
        dt_burst = time.perf_counter() - t0
        tim_b = dec.timing()
        dt_burst, _ = shard.reduce_timing(dt_burst, nfr_b, world, J.coll, cdev)
        reps = 1
        if args.sustain > 0:
            reps = max(1, int(np.ceil(args.sustain / max(dt_burst, 1e-6))))
            if world > 1 or J.coll is not None:           # every rank must use the same reps
                reps = int(shard.reduce_timing(float(reps), 0.0, world, J.coll, cdev)[0])
        clk0 = gpu_clock_mhz(J.local)
        barrier()
        t0 = time.perf_counter()
        nfr = 0
        marks = [t0]
        clk_mid, tele_mid = None, None
        for k_ in range(steps):
            for _ in range(reps):
                nfr += step()
            marks.append(time.perf_counter())
            if k_ == steps // 2:
                clk_mid = gpu_clock_mhz(J.local)
                tele_mid = gpu_telemetry(J.local)
        clk1 = gpu_clock_mhz(J.local)              # the last call(s) are still running
        nfr += drain()                             # the last call's results: all K steps fully delivered inside the timed region
        barrier()
        dt_local = time.perf_counter() - t0
        tim = dec.timing()
        step_ms = sorted((b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:]))
        dt, nfr_total = shard.reduce_timing(dt_local, nfr, world, J.coll, cdev)
        per_rank = shard.gather_scalars(dt_local, world, J.coll, cdev)
        return dict(warm=warm, tim_b=tim_b, dt_burst=dt_burst, reps=reps, clk0=clk0, clk_mid=clk_mid, clk1=clk1, tele_mid=tele_mid,
                    dt_local=dt_local, tim=tim, step_ms=step_ms, dt=dt, nfr_total=nfr_total, per_rank=per_rank)

    T = timed_region()
    warm, tim_b, dt_burst, reps, clk0, clk_mid, clk1, tele_mid = (T[k] for k in ('warm', 'tim_b', 'dt_burst', 'reps', 'clk0', 'clk_mid', 'clk1', 'tele_mid'))
    dt_local, tim, step_ms, dt, nfr_total, per_rank = (T[k] for k in ('dt_local', 'tim', 'step_ms', 'dt', 'nfr_total', 'per_rank'))
    # measurement aids (--ab, --decoders: same-process A/B of a per-launch switch, further decoders in the same process); not part
    # of the reported value, dead in the default run, and kept out of this function (the timed path must be auditable)
    ab = _probe_ab(args, J, step, drain, steps, nch, nout, M) if (args.ab and world == 1) else None
    trials = (_probe_decoders(args, J, make_decoder, step, drain, steps, reps, dt_local, nch, nout, M, dev)
              if (args.decoders > 1 and world == 1) else None)
    dec.close()
    if rank != 0:
        return None

    samples_per_pass = nch * nout * M                               # complex input samples per GPU per pass over the batch
    samples_per_step = samples_per_pass * reps
    value = world * samples_per_step * steps / dt / 1e6             # channel * Msamples/s
    # algorithmic bytes: 2 B per input sample per channel read (bps for the other formats), 4 B per
    # 12.5 kHz output written, taps (8 B each) read once per launch.  A step is `lps` pipelined FIR launches.
    lps = max(1, round(tim["fir_launches"] / (steps * reps)))       # launches per PASS
    pass_bytes = nstreams * nout * bps * M + nch * nout * 4 + lps * nch * ntaps * 8      # shared-stream mode: a stream's bytes count once
    step_bytes = pass_bytes * reps
    fir_bytes = pass_bytes / lps
    fir_avg_ms = tim["fir_ms"] / max(1, tim["fir_launches"])
    achieved = fir_bytes / (fir_avg_ms * 1e-3) / 1e9
    fir_ms_step = tim["fir_ms"] / steps
    burst = {"value": round(world * samples_per_pass * steps / dt_burst / 1e6, 1), "ms_per_pass": round(dt_burst / steps * 1e3, 4),
             "timed_region_s": round(dt_burst, 4), "whole_job_frac_of_hbm": round(pass_bytes * steps / dt_burst / 1e9 / HBM_PEAK_GBS, 4),
             "roofline_frac": round(fir_bytes / (tim_b["fir_ms"] / max(1, tim_b["fir_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "note": "`steps` single passes over the batch from a cold-ish device, as rounds 1-2 timed them; not the reported value"}
    msk_ms_step = warm["msk_ms"] / (warmup + (2 if repair else 1)) * reps      # (the gate's one or two passes are in the sum)
    if fmt == 0:
        kname = "fir_u8_shared_kernel" if share > 1 else J.fir_kernel_name(M, nout)
    else:
        # (mirrors acg_launch_fir_fmt: the wave-private kernel <FMT, 16-byte chunks per window (per plane), windows per tile> where it is
        #  instantiated for the window length, else round 1's workgroup-granular kernel)
        fid = {"cs16": 1, "split16": 2, "f32": 3}[fmt_name]
        shape = {("cs16", 160): (40, 32), ("cs16", 192): (48, 32), ("cs16", 200): (50, 32), ("f32", 200): (50, 32), ("f32", 240): (60, 16),
                 ("f32", 480): (120, 8), ("f32", 800): (200, 8), ("split16", 160): (20, 64)}.get((fmt_name, M))
        kname = ("fir_fmt_direct_kernel<%d, %d, %d>" % ((fid,) + shape)) if shape else "fir_fmt_kernel<%d>" % fid
    # HBM traffic of this launch shape from the committed PMC passes (rocprofv3 cannot run inside the timed
    # process): looked up by the full kernel signature and launch shape, not measured in this run -- the source is named next to the number
    traffic, traffic_src = lookup_traffic(kname, nch, M, ntaps, nblk / lps) if share == 1 else (None, None)
    whole = step_bytes * steps / dt / 1e9                            # per GPU
    out = {
        "value": round(value, 1),
        "ms_per_step": round(dt / steps * 1e3, 4),
        "timed_region_s": round(dt, 4),
        "sustain": {"passes_per_step": reps, "step_ms_min_median_max": [round(step_ms[0], 3), round(step_ms[len(step_ms) // 2], 3), round(step_ms[-1], 3)],
                    "shader_clock_mhz_start_mid_end": [clk0, clk_mid, clk1], "telemetry_mid_run": tele_mid,
                    "note": "a step = passes_per_step passes over the resident batch (chosen from the burst rate so that the timed region lasts "
                            ">= --sustain seconds); step times from host time stamps at the step boundaries (the host runs at most one call ahead "
                            "of the device); clocks from sysfs while the device is busy (null where the box does not expose them)"},
        "burst": burst,
        "data": "synthetic: " + data_desc,
        "config": {"workload": "%s: %d channels/GPU x %.1f Msps %s, one stream per channel, rtlMult=%d, ntaps=%d; step = %d pass(es) over a resident batch of "
                               "%d callbacks/channel in calls of %d; FIR decimate + MSK demod + framing%s, delivered to the host %d call(s) behind"
                               % (case["tag"], nch, 12500 * M / 1e6, {"u8": "u8 IQ", "cs16": "CS16 IQ", "split16": "split int16 I/Q", "f32": "real f32"}[fmt_name],
                                  M, ntaps, reps, nblk, cb, " + block repair + message split" if repair else "", args.collect_lag),
                   "signal_seconds_per_pass": round(nblk * 0.08192, 3),
                   "callbacks_per_call": cb, "collect_lag": args.collect_lag, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "case": name, "input_format": fmt_name, "channels_per_gpu": nch, "decim": M, "ntaps": ntaps, "blocks_per_step": nblk * reps, "blocks_per_pass": nblk, "passes_per_step": reps,
                   "input_bytes_per_gpu": int(nstreams * row),
                   "realtime_channels_equiv": int(value / (12500 * M / 1e6)),
                   "arithmetic": "%s in, f32 down-converter and matched filter, f64 VCO/PLL/normalisation (as the reference)"
                                 % {"u8": "u8 I/Q", "cs16": "int16 I/Q", "split16": "split int16 I/Q", "f32": "real f32"}[fmt_name],
                   "delivered": ("acg_msg records: blocks checked / repaired on the device (ACG_F_REPAIR, acars.c:93-215) and split into outputmsg()'s fields "
                                 "(output.c:486-560) by acg_collect_msgs, inside the timed region" if repair else
                                 "pre-repair blocks (acg_collect_frames, --raw-blocks)"),
                   "channels_total": nch_total, "blocks_decoded_timed": int(nfr_total),
                   "contexts": "one context from acg_create, as a host gets it (no placement selection)" if not trial_ms or trial_best == 0 else "best of %d contexts (--placement-keep best)" % len(trial_ms),
                   "placement": ({"contexts_tried": len(trial_ms), "ms_per_call": [round(x, 3) for x in trial_ms], "kept": trial_best,
                                  "spread": round(max(trial_ms) / min(trial_ms) - 1.0, 4),
                                  "fir_ms_per_launch": ([round(x, 4) for x in D.best_placed.last_fir_ms] if getattr(D.best_placed, "last_fir_ms", None) else None),
                                  "note": "diagnostic (--placements N), untimed: N contexts alive at once, acg_placement_trial on each after a warm-up round; "
                                          "`kept` is the one timed (0 = the first, unless --placement-keep best)"}
                                 if trial_ms else None)},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": (traffic_src + " (rocprofv3 PMC passes of the same launch shape: 2 x FETCH_SIZE + WRITE_SIZE; "
                                        "looked up by the full kernel signature, not collected in this run)") if traffic else None,
                     "bytes_per_launch": int(fir_bytes), "avg_launch_ms": round(fir_avg_ms, 4), "launches_per_step": lps * reps,
                     "launches_per_pass": lps,
                     "timing": "HIP events around every launch of the kernel on its own stream, inside the timed region "
                               "(the demodulator of the previous call / chunk runs beside it)",
                     "frac_of_measured_copy_ceiling_6290": round(achieved / COPY_CEILING_GBS, 4)},
        "whole_job_frac_of_hbm": round(whole / HBM_PEAK_GBS, 4),
        "whole_job_GBs_per_gpu": round(whole, 1),
        "time_dominant_kernel": "msk_demod_kernel" if msk_ms_step > fir_ms_step else kname,
        "kernels": {"fir_ms_per_step": round(fir_ms_step, 4), "msk_ms_per_step": round(msk_ms_step, 4),
                    "note": "per-step sums of event-timed launches; the stages overlap (down-converter of call/chunk i+1 beside the "
                            "demodulator of i); the demodulator figure is taken during warm-up (its events are off in the timed region)"},
        "parity": parity,
    }
    if ab:
        out["ab_same_process"] = ab
    if trials:
        out["placement_trials"] = trials
    if ntaps != M:
        out["config"]["filter"] = ("%d-tap low-pass = the channel's NCO taps (rtl.c:283-286) x Hamming window, unit DC gain; the reference "
                                   "only has the boxcar, so the oracle for this filter is the same sum(vb*wf) formula with these taps" % ntaps)
    if world > 1:
        out["per_gpu"] = [round(samples_per_step * steps / t / 1e6, 1) for t in per_rank]
    if J.coll is not None and world == 1:
        out["config"]["collectives"] = "forced through torch.distributed/%s with world size 1 (--rccl-selftest)" % J.backend
    if share > 1:
        out["config"]["channels_per_stream"] = share
        out["roofline"]["note"] = ("shared-stream mode: %d channels reuse each stream's bytes, the down-converter is VALU-bound "
                                   "(8*K flop per 2 B); achieved counts each stream once and is NOT the HBM roofline figure" % share)
        keff = min(share, 8)
        ops = nch * nout * M * (2.0 + 3.0 / keff) * steps / (tim["fir_ms"] * 1e-3)
        out["valu"] = {"kernel": "fir_u8_shared_kernel", "lane_ops_per_s": round(ops, 0), "peak": 256 * 4 * 16 * 2.4e9,
                       "frac": round(ops / (256 * 4 * 16 * 2.4e9), 4), "lane_ops_per_channel_sample": round(2.0 + 3.0 / keff, 3)}
    return out


def run_hostfed(J, args, steps, warmup):
    """acg_process_iq_u8_host at north-star width from pinned host memory: every call hands the library a host buffer
    (two alternate, like a driver's ring) that is free again when the call returns; the library copies it to one of two device
    staging buffers beside the kernels of the previous call.  Reports channel*Msps, the fraction of this box's measured
    host-to-device rate, and whether 10 000 channels x 2.5 Msps (25 000 channel*Msps = 50 GB/s) is sustained.
    Gate: the delivered records of a pass from reset equal those of the _dev entry point on the same bytes (all channels),
    and the first channels' blocks equal the oracle's demodulator + repair on the GPU's dm."""
    import ctypes as C
    import numpy as np
    import torch
    from acarsdec_amd import decoder as D, synth as S, _capi as K
    L, dev = J.L, J.dev
    nch, M, ntaps, cb = HOSTFED["channels"], HOSTFED["decim"], HOSTFED["ntaps"], HOSTFED["call_blocks"]
    if args.hostfed_channels:
        nch = args.hostfed_channels
    nbuf = 2
    row = cb * 1024 * M * 2
    nout = nbuf * cb * 1024
    r0 = np.random.default_rng(0xACA25 + 7)
    offs = r0.integers(-48, 49, size=nch) * 25000.0
    offs[np.abs(offs) < 25000] = 50000.0
    phases = r0.uniform(0, 2 * np.pi, nch)
    taps = make_taps(D, "u8", offs, M, ntaps)
    # content on the device first (the up-converter is a device kernel): random bytes everywhere, ACARS traffic on the gate's channels
    ncheck = min(args.check_channels, nch)
    full = J.iq_all[: nch * nbuf * row].view(nch, nbuf * row)
    assert L.acg_fill_random_u8_dev(full.data_ptr(), nbuf * row, nch, nbuf * row, 0xACA25 + 99, None) == 0
    sigma = SCALE * CARRIER * (M / (2.0 * 10 ** (SNR_DB / 10.0))) ** 0.5
    trk = np.empty((ncheck, nout), dtype=np.float32)
    for c in range(ncheck):
        a, _ = S.channel_audio(np.random.default_rng(0xACA25 + 5000 + c), nout, gap=(500, 1500), text_len=(5, 40))
        trk[c] = CARRIER * (1.0 + DEPTH * a)
    d_trk = torch.from_numpy(trk).to(dev)
    # (named tensors: a temporary's memory goes back to the caching allocator the moment data_ptr() has been taken, and the next
    #  temporary gets the same address -- round 4's first version of this function handed the kernel three aliases of one buffer)
    d_idx = torch.arange(ncheck, dtype=torch.int32, device=dev)
    d_off = torch.from_numpy(offs[:ncheck].astype(np.float32)).to(dev)
    d_ph = torch.from_numpy(phases[:ncheck].astype(np.float32)).to(dev)
    assert L.acg_synth_iq_u8_dev(full.data_ptr(), nbuf * row, ncheck, nout, M, d_trk.data_ptr(), nout, d_idx.data_ptr(),
                                 d_off.data_ptr(), d_ph.data_ptr(), SCALE, sigma, 0xACA25, None) == 0
    torch.cuda.synchronize()
    del d_trk, d_idx, d_off, d_ph
    # the host side: nbuf pinned buffers of one call each
    t_pin = time.perf_counter()
    hptr = [L.acg_host_alloc(nch * row) for _ in range(nbuf)]
    assert all(hptr), "acg_host_alloc failed"
    hview = [np.ctypeslib.as_array(C.cast(p_, C.POINTER(C.c_ubyte)), shape=(nch, row)) for p_ in hptr]
    for b in range(nbuf):
        torch.from_numpy(hview[b]).copy_(full[:, b * row:(b + 1) * row])
    torch.cuda.synchronize()
    pin_s = time.perf_counter() - t_pin
    gbs = C.c_double(0)
    scratch = J.iq_all[nch * nbuf * row: nch * nbuf * row + nch * row]
    assert L.acg_probe_h2d(scratch.data_ptr(), hptr[0], nch * row, 3, C.byref(gbs)) == 0
    h2d = gbs.value
    maxm = nch * 4 + 8192

    def mk():
        d_ = D.Decoder(nch, decim=M, ntaps=ntaps, max_blocks=cb, device=J.local, bitlog=False, timing=True, repair=True, max_lag=1)
        d_.set_taps(taps)
        return d_
    dec = mk()

    def call(b, sink=None, lag=1):
        rc = L.acg_process_iq_u8_host(dec.ctx, hptr[b], row, cb)
        if rc != 0:
            raise K.AcgError(rc, L.acg_last_error(dec.ctx).decode())
        n = 0
        while True:
            m, fb, more = dec.collect_msgs_raw(lag, maxm)
            if sink is not None:
                sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(m)]
            n += m
            if not more:
                return n

    def drain(sink=None):
        n = 0
        while True:
            m, fb, more = dec.drain_msgs_raw(maxm)
            if sink is not None:
                sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(m)]
            n += m
            if not more:
                return n
    # ---- gate
    from oracle import oracle as O
    got = []
    dm_gpu = [[] for _ in range(ncheck)]
    for b in range(nbuf):
        call(b, got, lag=0)
        for c in range(ncheck):
            dm_gpu[c].append(dec.dm(c, cb * 1024))
    drain(got)
    ref = mk()                                   # the _dev entry point on the same bytes
    want = []
    for b in range(nbuf):
        ref.in_callback(full[:, b * row:(b + 1) * row], nblocks=cb, pitch=nbuf * row)
    want = ref.drain_msgs(maxm)
    ref.close()
    key = lambda m_: (int(m_.chn), int(m_.end_bit))
    same_as_dev = sorted(bytes(m_) for m_ in got) == sorted(bytes(m_) for m_ in want)
    per = {}
    for m_ in sorted(got, key=key):
        per.setdefault(int(m_.chn), []).append(O.msg_tuple(m_))
    ok, nblocks = True, 0
    for c in range(ncheck):
        ch = O.Channel(c)
        ch.demod(np.concatenate(dm_gpu[c]))
        kept = [b_ for b_ in (O.blk_process(f) for f in ch.frames) if b_ is not None]
        nblocks += len(kept)
        ok &= per.get(c, []) == [O.msg_tuple(O.msg_split(b_)) for b_ in kept]
    parity = dict(channels_checked=ncheck, blocks=nblocks, blocks_exact_given_gpu_dm=bool(ok), dm_within_1e5_rel=True,
                  msgs=dict(records=nblocks, exact=bool(ok), delivered=len(got)),
                  same_records_as_dev_entry_point=bool(same_as_dev), records_all_channels=len(got),
                  end_to_end=dict(blocks_differing=0, allowed=0, note="the down-converter kernel is the _dev path's; this gate is about the host feed"))
    if not (ok and same_as_dev and nblocks > 0):
        raise SystemExit("bench[hostfed]: host-fed output differs: %r" % parity)
    # ---- timing
    for _ in range(max(1, warmup)):
        for b in range(nbuf):
            call(b)
    drain()
    dec.timing()
    dec.set_timing(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(nbuf):
        call(b)
    drain()
    torch.cuda.synchronize()
    per_pair = time.perf_counter() - t0
    reps = max(1, int(np.ceil(args.sustain / max(per_pair * steps, 1e-6)))) if args.sustain > 0 else 1
    dec.timing()                       # (the probe's launches are nobody's roofline: the event sums start with the timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nrec = 0
    marks = [t0]
    for _ in range(steps):
        for _ in range(reps):
            for b in range(nbuf):
                nrec += call(b)
        marks.append(time.perf_counter())
    nrec += drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tim = dec.timing()
    dec.close()
    for p_ in hptr:
        L.acg_host_free(p_)
    ncalls = steps * reps * nbuf
    value = nch * cb * 1024 * M * ncalls / dt / 1e6
    in_gbs = nch * row * ncalls / dt / 1e9
    fir_bytes = nch * cb * 1024 * (2 * M + 4) + nch * ntaps * 8
    fir_avg_ms = tim["fir_ms"] / max(1, tim["fir_launches"])
    achieved = fir_bytes / (fir_avg_ms * 1e-3) / 1e9 * (ncalls / max(1, tim["fir_launches"]))
    step_ms = sorted((b_ - a_) * 1e3 for a_, b_ in zip(marks[:-1], marks[1:]))
    need = nch * 12500 * M / 1e6
    return {
        "value": round(value, 1), "ms_per_step": round(dt / steps * 1e3, 4), "timed_region_s": round(dt, 4),
        "sustain": {"passes_per_step": reps, "step_ms_min_median_max": [round(step_ms[0], 3), round(step_ms[len(step_ms) // 2], 3), round(step_ms[-1], 3)]},
        "whole_job_frac_of_hbm": round(fir_bytes * ncalls / dt / 1e9 / HBM_PEAK_GBS, 4),
        "time_dominant_kernel": "host-to-device copy (PCIe)",
        "hostfed": {"input_GBs": round(in_gbs, 2), "h2d_GBs_measured": round(h2d, 2), "frac_of_h2d": round(in_gbs / h2d, 4),
                    "realtime_needs": need, "realtime": bool(value >= need), "pin_and_fill_s": round(pin_s, 2)},
        "roofline": {"bound": "hbm", "kernel": J.fir_kernel_name(M, cb * 1024), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": lookup_traffic(J.fir_kernel_name(M, cb * 1024), nch, M, ntaps, cb * ncalls / max(1, tim["fir_launches"]))[0],
                     "bytes_per_launch": int(fir_bytes * ncalls / max(1, tim["fir_launches"])),
                     "launches_per_pass": max(1, tim["fir_launches"] // max(1, steps * reps)),
                     "avg_launch_ms": round(fir_avg_ms, 4), "launches_per_step": tim["fir_launches"] // steps,
                     "note": "the kernel's own launches (event-timed) while the NEXT call's host-to-device copy runs beside them; the job is bound by the link, not by this kernel"},
        "parity": parity,
        "data": "synthetic: uniform random bytes per stream, ACARS/MSK traffic (AM depth %.1f, %.0f dB SNR) on the %d gate channels; generated on the device, "
                "copied once into %d pinned host buffers (acg_host_alloc) of one call each" % (DEPTH, SNR_DB, ncheck, nbuf),
        "config": {"workload": "%s: %d channels x %.1f Msps u8 IQ handed over from pinned HOST memory in calls of %d callbacks (%.1f GB per call, two buffers "
                               "alternating), acg_process_iq_u8_host (buffer free on return) + acg_collect_msgs one call behind; step = %d x %d calls"
                               % (HOSTFED["tag"], nch, 12500 * M / 1e6, cb, nch * row / 1e9, reps, nbuf),
                   "case": "hostfed", "input_format": "u8", "channels_per_gpu": nch, "decim": M, "ntaps": ntaps, "callbacks_per_call": cb, "passes_per_step": reps,
                   "blocks_per_pass": nbuf * cb, "blocks_per_step": nbuf * cb * reps,
                   "arithmetic": "u8 I/Q in, f32 down-converter and matched filter, f64 VCO/PLL/normalisation (as the reference)",
                   "delivered": "acg_msg records (ACG_F_REPAIR + acg_collect_msgs), inside the timed region",
                   "contexts": "one context from acg_create, as a host gets it (no placement selection)",
                   "records_delivered_timed": int(nrec)},
    }


MULTI_GPU_NOTE = ("no N>1 run has been measured by the builder (1-GPU boxes only): --gpus N shards channel c to rank c mod N "
                  "(weak scaling, no data-path collective; RCCL carries the 32 B/channel config-table broadcast, barriers and reductions)")


# ------------------------------------------------------------------------------------------ BASELINE configs[1]: the rtl.c shape
RTL8 = dict(tag="BASELINE configs[1]: one dongle, 8 channels on ONE 2.0 Msps u8 stream (rtl.c's shape)", decim=160, callbacks=32)


def rtl8_cpu_child(variant, nfreq, path):
    """child process: the UNMODIFIED reference (oracle/_ref, its own flags) -- initRtl for the dongle's channels, then in_callback
    (rtl.c:314-361: mix + decimate for all channels, demodMSK per channel, decodeAcars) over the file's callbacks on one core"""
    import numpy as np
    from oracle import oracle as O
    M = RTL8["decim"]
    freqs = rtl8_freqs(int(nfreq))
    os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
    ref = O.Ref(variant)
    ref.init_rtl(freqs, M)
    iq = np.fromfile(path, dtype=np.uint8)
    blk = 1024 * M * 2
    bufs = [np.ascontiguousarray(iq[b * blk:(b + 1) * blk]) for b in range(iq.size // blk)]
    for b in bufs:                      # warm-up: one pass over the file (page faults, caches, the core's clock)
        ref.in_callback(b)
    ref.init_rtl(freqs, M)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.5:   # whole passes over the file for >= 1.5 s (a single 32-callback pass is 10-80 ms: too short to time)
        for b in bufs:
            ref.in_callback(b)
        n += len(bufs)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(ms_per_callback=dt / n * 1e3, callbacks=n)))


def rtl8_freqs(nch):
    return ["%.3f" % (131.025 + 0.050 * k) for k in range(nch)]


def rtl8_oneline(chn, lvl, err, addr, fid, mode, label, no, txt):
    """printoneline() (output.c:327-346) without the date"""
    t = txt.split(b"\0")[0][:59].replace(b"\n", b" ").replace(b"\r", b" ")
    dec = lambda b: b.split(b"\0")[0].decode("latin-1")
    return "#%1d (L:%+5.1f E:%1d) %7s %6s %1s %2s %4s %s" % (chn + 1, lvl, err, dec(addr), dec(fid), dec(mode) or "\0", dec(label), dec(no), t.decode("latin-1"))


def run_rtl8(J, args):
    """BASELINE configs[1] and the path the north star names: nbch channels of ONE dongle on one 2.0 Msps u8 I/Q stream, handed
    over from host memory one reference callback (1024 outputs = 81.92 ms of signal, rtl.c:49,213) at a time.
      legacy   the reference's UNCHANGED acarsdec.c + acars.c + output.c + rtl.c with the one-hunk binding on
               compat_msk.c: acarsdec_amd_in_callback -> GPU -> every bit replayed through the unchanged decodeAcars() on the
               reference's own channel[] (lib/acarsdec_gpu_rtl, a file-playing librtlsdr stand-in); time inside the entry point
      batched  the same bytes through acg_process_iq_u8_host (nstreams = 1) + acg_collect_msgs one call behind
      cpu      the unmodified reference's in_callback on the same bytes, its own flags, one core (oracle/_ref)
    Parity: the legacy program's printed messages == the CPU twin program's (oracle/_ref/acarsdec_cpu_rtl), and the batched
    API's records, printed the same way, == both."""
    import re
    import tempfile
    import numpy as np
    from acarsdec_amd import decoder as D, synth as S, _capi as K
    M, ncb = RTL8["decim"], RTL8["callbacks"]
    gpu_exe = os.path.join(ROOT, "acarsdec_amd", "lib", "acarsdec_gpu_rtl")
    cpu_exe = os.path.join(ROOT, "oracle", "_ref", "acarsdec_cpu_rtl")
    out = {"workload": RTL8["tag"] + "; %d callbacks of 1024 outputs from host memory, rtlMult=%d" % (ncb, M), "budget_ms_per_callback": 81.92,
           "decim": M, "callbacks": ncb}
    strip = lambda txt: [l for l in re.sub(r"\d\d/\d\d/\d{4} \d\d:\d\d:\d\d\.\d{3} ", "", txt).splitlines() if l.startswith("#")]
    per_ch = lambda lines: {k: [l for l in lines if l.split()[0] == k] for k in sorted(set(l.split()[0] for l in lines))}
    with tempfile.TemporaryDirectory() as td:
        for nch in (8, 16):
            rng = np.random.default_rng(0x0881 + nch)
            freqs = rtl8_freqs(nch)
            fr = [D.parse_freq_mhz(f) for f in freqs]
            fc, _ = D.choose_fc(fr, M)
            env = np.zeros((nch, ncb * 1024))
            for c in range(nch):
                a_, _ = S.channel_audio(rng, env.shape[1], gap=(3125, 12500), text_len=(20, 120))
                env[c] = CARRIER * (1.0 + DEPTH * a_)
            iq = S.iq_u8_from_envelopes(env, M, [f - fc for f in fr], phases=list(rng.uniform(0, 2 * np.pi, nch)), scale=1.0 / nch, noise=0.004, rng=rng)
            path = os.path.join(td, "rtl%d.iq" % nch)
            iq.tofile(path)
            e = {"channels": nch}
            envp = dict(os.environ, ACARSDEC_IQ_FILE=path, ACARSDEC_AMD_STATS="1")
            lines = {}
            for name, exe in (("cpu", cpu_exe), ("legacy", gpu_exe)):
                if not os.path.exists(exe):
                    continue
                r = subprocess.run([exe, "-o", "1", "-r", "0"] + freqs, env=envp, capture_output=True, timeout=300)
                if r.returncode != 0:
                    e[name + "_error"] = _short(r.stderr.decode("latin-1"), 200)
                    continue
                lines[name] = per_ch(strip(r.stdout.decode("latin-1")))
                if name == "legacy":
                    m = re.search(r"first call ([0-9.]+) ms.*others ([0-9.]+) ms per call", r.stderr.decode("latin-1"))
                    if m:
                        e["legacy_first_call_ms"], e["legacy_ms_per_callback"] = float(m.group(1)), float(m.group(2))
            # the batched API, nstreams = 1: one host buffer per callback, messages collected one call behind
            dec = D.Decoder(nch, decim=M, nstreams=1, max_blocks=1, repair=True, bitlog=False, max_lag=1)
            dec.init_rtl(freqs)
            blk = 1024 * M * 2
            bufs = [np.ascontiguousarray(iq[b * blk:(b + 1) * blk]).reshape(1, -1) for b in range(ncb)]

            def run_batched(sink):
                for b in bufs:
                    dec.in_callback(b)
                    while True:
                        n_, fb, more = dec.collect_msgs_raw(1, 256)
                        if sink is not None:
                            sink += [K.Msg.from_buffer_copy(fb[i]) for i in range(n_)]
                        if not more:
                            break
                last = dec.drain_msgs(256)
                if sink is not None:
                    sink += last
            run_batched(None)                       # warm-up (first launches), then from reset
            dec.reset()
            msgs = []
            t0 = time.perf_counter()
            run_batched(msgs)
            e["batched_ms_per_callback"] = round((time.perf_counter() - t0) / ncb * 1e3, 4)
            dec.close()
            got = per_ch([rtl8_oneline(int(m.chn), m.lvl, int(m.err), m.addr, m.fid, m.mode, m.label, m.no, bytes(m.txt[: m.txt_len])) for m in msgs])
            lines["batched"] = got
            # the reference's in_callback on one host core (its own flags)
            for variant in ("_fast", "_v3", ""):
                if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libacarsref%s.so" % variant)):
                    continue
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rtl8-cpu-child", variant, str(nch), path], capture_output=True, text=True, timeout=300)
                if r.returncode == 0 and r.stdout.strip():
                    e["cpu_reference_ms_per_callback"] = round(json.loads(r.stdout.strip().splitlines()[-1])["ms_per_callback"], 4)
                    break
            e["messages"] = sum(len(v) for v in lines.get("batched", {}).values())
            e["parity"] = {"legacy_program_equals_cpu_program": (lines["legacy"] == lines["cpu"]) if ("legacy" in lines and "cpu" in lines) else None,
                           "batched_equals_cpu_program": (lines["batched"] == lines["cpu"]) if "cpu" in lines else None,
                           "batched_equals_legacy_program": (lines["batched"] == lines["legacy"]) if "legacy" in lines else None}
            if any(v is False for v in e["parity"].values()):
                raise SystemExit("bench[rtl%d]: printed messages differ: %r" % (nch, {k: {c: len(v) for c, v in l.items()} for k, l in lines.items()}))
            out["ch%d" % nch] = e
    return out


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_line(full):
    """The driver-facing result line (< 4 KB) out of the full detail dict: every key the contract names, the roofline of the
    dominant kernel, the CPU baseline, the parity verdicts, and a five-number summary per "also" case.  Pure function (a CPU
    test feeds it a worst-case detail and measures the line)."""
    def parity_short(p):
        if not p:
            return None
        refs = p.get("reference_builds") or {}
        ex = p.get("exact_order_mode") or {}
        ms = p.get("msgs") or {}
        return {"channels": p.get("channels_checked"), "blocks": p.get("blocks"), "exact_given_gpu_dm": p.get("blocks_exact_given_gpu_dm"),
                "msgs": ms.get("records"), "msgs_exact": ms.get("exact"), "dm_within_1e5_rel": p.get("dm_within_1e5_rel"),
                "exact_order_identical": (bool(ex.get("dm_bit_identical_to_oracle") and ex.get("blocks_identical_end_to_end")) if ex else None),
                "end_to_end_differing": (p.get("end_to_end") or {}).get("blocks_differing"), "allowed": (p.get("end_to_end") or {}).get("allowed"),
                "ref_builds_differing": refs.get("ref_fast_vs_ref_o2_blocks_differing"),
                "gpu_vs_ref_ofast": refs.get("gpu_vs_ref_ofast_blocks_differing")}

    def parity_ok(p):
        if not p:
            return None
        e = p.get("end_to_end") or {}
        ms = p.get("msgs")
        return bool(p.get("blocks_exact_given_gpu_dm") and p.get("dm_within_1e5_rel") and (ms is None or ms.get("exact"))
                    and (e.get("allowed") is None or e.get("blocks_differing", 0) <= e["allowed"]) and not e.get("gpu_vs_ref_ofast"))
    cfg = full.get("config", {})
    rf = full.get("roofline", {})
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype")}
    line["data"] = _short(full.get("data", "synthetic"), 160)
    line["config"] = {"workload": _short(cfg.get("workload", ""), 300), "case": cfg.get("case"), "channels_per_gpu": cfg.get("channels_per_gpu"),
                      "decim": cfg.get("decim"), "ntaps": cfg.get("ntaps"), "callbacks_per_call": cfg.get("callbacks_per_call"), "collect_lag": cfg.get("collect_lag"),
                      "passes_per_step": cfg.get("passes_per_step"), "input_format": cfg.get("input_format"),
                      "delivered": _short(cfg.get("delivered", ""), 60), "contexts": _short(cfg.get("contexts", ""), 60)}
    if cfg.get("placement"):
        line["config"]["placement_ms_per_call"] = cfg["placement"].get("ms_per_call")
    line["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_live", "bytes_per_launch", "avg_launch_ms",
                                                "launches_per_step", "pure_reader_GBs_measured_this_run")}
    for k in ("whole_job_frac_of_hbm", "time_dominant_kernel", "timed_region_s", "per_gpu"):
        if k in full:
            line[k] = full[k]
    line["parity"] = parity_short(full.get("parity"))
    if full.get("also"):
        line["also"] = {}
        for name, a in full["also"].items():
            if "error" in a:
                line["also"][name] = {"error": _short(a["error"], 120)}
                continue
            if name == "rtl8":                       # BASELINE configs[1]: ms per 81.92 ms callback, legacy view / batched API / CPU reference
                line["also"][name] = {"budget_ms": a.get("budget_ms_per_callback")}
                for k in ("ch8", "ch16"):
                    c_ = a.get(k) or {}
                    pv = [v for v in (c_.get("parity") or {}).values() if v is not None]
                    line["also"][name][k] = {"legacy_ms": c_.get("legacy_ms_per_callback"), "batched_ms": c_.get("batched_ms_per_callback"),
                                             "cpu_ref_ms": c_.get("cpu_reference_ms_per_callback"), "msgs": c_.get("messages"),
                                             "parity_ok": (all(pv) if pv else None)}
                continue
            ar = a.get("roofline", {})
            e = {"value": a.get("value"), "ms_per_step": a.get("ms_per_step"), "channels": a.get("config", {}).get("channels_per_gpu"),
                 "whole_job_frac": a.get("whole_job_frac_of_hbm"), "roofline_frac": ar.get("frac"), "traffic": ar.get("traffic"),
                 "bytes_per_launch": ar.get("bytes_per_launch"), "parity_ok": parity_ok(a.get("parity")),
                 "blocks": (a.get("parity") or {}).get("blocks"), "e2e_differing": ((a.get("parity") or {}).get("end_to_end") or {}).get("blocks_differing"),
                 "gpu_vs_ref_ofast": ((a.get("parity") or {}).get("end_to_end") or {}).get("gpu_vs_ref_ofast")}
            if a.get("config", {}).get("placement"):
                e["placement_ms_per_call"] = a["config"]["placement"].get("ms_per_call")
            if "hostfed" in a:
                e["hostfed"] = a["hostfed"]
            if "per_gpu" in a:
                e["per_gpu"] = a["per_gpu"]
            line["also"][name] = e
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample", ""), 120), "all_cores": cb.get("all_cores"), "gpu_over_cpu": cb.get("gpu_over_cpu")}
    line["multi_gpu"] = _short(full.get("multi_gpu", ""), 120)
    line["detail"] = "bench_detail.json / the '# bench_detail:' stdout line"
    # the budget is enforced, not hoped for: optional keys go, least important first, until the line fits
    size = lambda: len(json.dumps(line, separators=(",", ":")))
    also = line.get("also", {})
    trims = ([lambda a=a: a.pop("placement_ms_per_call", None) for a in also.values()] +
             [lambda a=a: a.pop("bytes_per_launch", None) for a in also.values()] +
             [lambda a=a: a.__setitem__("per_gpu", [int(round(x)) for x in a["per_gpu"]]) if "per_gpu" in a else None for a in also.values()] +
             [lambda: line.__setitem__("data", _short(line["data"], 60)),
              lambda: line["config"].__setitem__("workload", _short(line["config"]["workload"], 160)),
              lambda: line.__setitem__("multi_gpu", _short(line["multi_gpu"], 60))] +
             [lambda a=a: a.pop("traffic", None) for a in also.values()] +
             [lambda a=a: a.pop("per_gpu", None) for a in also.values()])
    for t in trims:
        if size() <= 3900:
            break
        t()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CASES), default="throughput", help="the case reported as `value`")
    ap.add_argument("--also", default=None,
                    help="comma-separated cases timed in the same invocation and reported under \"also\" (default: wide,stress on "
                         "one GPU; shard2048 on several; 'none' to skip)")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (overrides the case; also: no \"also\" cases)")
    ap.add_argument("--decim", type=int, default=None, help="rtlMult: 200 = 2.5 Msps")
    ap.add_argument("--ntaps", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None, help="1024-output callbacks per channel per step")
    ap.add_argument("--call-blocks", type=int, default=8, help="callbacks handed to the library per call (a step streams its batch in calls)")
    ap.add_argument("--check-channels", type=int, default=64, help="channels of rank 0 verified against the oracle (SURVEY 8d: 64)")
    ap.add_argument("--format", choices=["u8", "cs16", "split16", "f32"], default="u8",
                    help="input sample format: u8 = rtl.c (headline); cs16 = soapy.c, split16 = sdrplay.c, f32 = air.c (SURVEY 8f.2)")
    ap.add_argument("--share", type=int, default=1,
                    help="channels per input stream (rtl.c's own shape: one dongle feeds up to 16 channels); >1 = shared-stream "
                         "mode, VALU-bound, reported separately and never as the roofline figure (SURVEY 8d)")
    ap.add_argument("--bitlog", type=int, default=1, help="1: the demodulator also writes its per-bit soft symbols (vo, level: 8 B per bit) to HBM")
    ap.add_argument("--placements", type=int, default=1,
                    help="diagnostic: N contexts alive at once, acg_placement_trial on each before the run (untimed); their times go to "
                         "config.placement.  The context that is timed is the FIRST one (what a host gets from acg_create) unless "
                         "--placement-keep best")
    ap.add_argument("--placement-keep", choices=["first", "best"], default="first")
    ap.add_argument("--collect-lag", type=int, default=2,
                    help="how many calls behind the newest the host collects results (acg_collect_msgs(lag)); the context's block queue is "
                         "sized for it (acg_config.max_lag).  2 (default): two calls in flight, the host never waits on the newest call's "
                         "predecessor; 1: classic double buffering (rounds 1-3)")
    ap.add_argument("--raw-blocks", action="store_true",
                    help="time the pre-repair blocks (acg_collect_frames without ACG_F_REPAIR) as rounds 1-3 did, instead of the "
                         "delivered acg_msg records")
    ap.add_argument("--hostfed-channels", type=int, default=None, help="channels of the hostfed case (default 10 000)")
    ap.add_argument("--hostfed-child", action="store_true",
                    help="(internal) run only the hostfed case and print its dict: the parent runs it in a child process with a "
                         "timeout AFTER its own cases, so that whatever pinning tens of GB of host memory does on a box cannot take "
                         "the result line with it")
    ap.add_argument("--detail-file", default=None, help="where the full per-case detail goes (default: bench_detail.json next to bench.py, "
                                                        "and gpurun_out/ when that exists)")
    ap.add_argument("--decoders", type=int, default=1, help="measurement aid: time this many decoders (separate allocations) in the same process")
    ap.add_argument("--ab", default=None, help="measurement aid: comma-separated ACG_FIR_VARIANT values (or NAME=v1,v2 for another per-launch "
                                               "switch, e.g. ACG_MSK_LPC_LIVE=2,4) timed after the run in the same process, same decoder")
    ap.add_argument("--sustain", type=float, default=5.0,
                    help="seconds the timed region of every case should last at least: a step becomes as many passes over the resident "
                         "batch as that takes (0 = one pass per step, the short burst of rounds 1-2)")
    ap.add_argument("--no-ref-leg", action="store_true", help="skip the gate's reference -O2 vs -Ofast leg (oracle/_ref on the host cores)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="with --gpus 1: initialise torch.distributed (nccl = RCCL) with world size 1 and send the channel scatter, the barriers "
                         "and the reductions through it on device tensors instead of the world == 1 short-cuts")
    ap.add_argument("--cooldown", type=float, default=3.0,
                    help="idle seconds between two cases of one invocation.  The cases that saturate HBM hold the chip at its 1400 W cap "
                         "(shader clock 1.64 GHz); a case that starts right behind one inherits that state for its first seconds, and "
                         "the demodulator-bound cases follow the shader clock (round 5: 2048 channels 0.57 of HBM right behind the "
                         "4096-channel case, 0.60-0.63 from an idle chip).  Every case is still timed for >= --sustain seconds.")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic of the headline launch shape in this invocation (two short child runs under "
                         "rocprofv3 --pmc, ~25 s each); the committed PMC passes of profiles/pmc_traffic.json are looked up instead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-child", nargs=4, default=None)
    ap.add_argument("--rtl8-cpu-child", nargs=3, default=None)
    args = ap.parse_args()
    if args.rtl8_cpu_child:
        rtl8_cpu_child(*args.rtl8_cpu_child)
        return
    if args.cpu_child:
        v, M, b, s = args.cpu_child
        cpu_baseline_child(v, int(M), int(b), float(s))
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from acarsdec_amd import _capi as K

    J = Job()
    J.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    J.rank = rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    J.backend = os.environ.get("ACG_BENCH_BACKEND", "nccl")    # "gloo": rehearsal without RCCL (several ranks on one GPU)
    if J.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, torch.cuda.device_count()))
    J.local = local = local % torch.cuda.device_count()       # (the gloo rehearsal maps all ranks of a 1-GPU box to GPU 0)
    torch.cuda.set_device(local)
    J.dev = dev = torch.device("cuda", local)
    J.cdev = dev if J.backend == "nccl" else None              # where the few collective tensors live
    J.dist = dist
    J.coll = None                                              # torch.distributed where collectives are used, else None
    if world > 1 or args.rccl_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if J.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(J.backend)
        J.coll = dist
    J.L = K.load()

    def fir_kernel_name(M, nout):
        # (the product library knows variant 5 and its fallback only: ACG_FIR_VARIANT means something to the lab build alone)
        v = int(os.environ.get("ACG_FIR_VARIANT", "5")) if J.L.acg_is_lab_build() else 5
        if (v in (7, 8) or 70 <= v <= 73) and M == 200 and nout % 128 == 0:
            return "fir_u8_coltap_kernel"
        if (v in (5, 7, 8) or 50 <= v <= 55 or 70 <= v <= 73) and M in (160, 192, 200) and nout % 128 == 0:
            return "fir_u8_direct_kernel<%d, 0, 0, true, %s, false>" % (M // 8, "false" if v == 55 else "true")     # (as rocprofv3 prints it)
        return {0: "fir_u8_tile_kernel", 4: "fir_u8_dma_kernel"}.get(v, "fir_u8_persist_kernel")
    J.fir_kernel_name = fir_kernel_name

    case = dict(CASES[args.config])
    overridden = any(x is not None for x in (args.channels, args.decim, args.ntaps, args.blocks))
    if args.channels:
        case["channels"] = args.channels
    if args.decim:
        case["decim"] = args.decim
        case["ntaps"] = args.decim
    if args.ntaps:
        case["ntaps"] = args.ntaps
    if args.blocks:
        case["blocks"] = args.blocks
    elif args.channels:
        case["blocks"] = 8
    if args.also is not None:
        also = [] if args.also in ("", "none") else args.also.split(",")
    elif overridden or args.format != "u8" or args.share > 1:
        also = []
    else:
        # (order: the two cases whose step the demodulator sets -- they follow the shader clock -- before the cases that saturate
        #  HBM and run the chip into its power cap: see --cooldown)
        also = ["shard2048", "wide", "stress", "cs16", "f32", "rtl8", "hostfed"] if world == 1 else ["shard2048"]
        also = [a for a in also if a != args.config]
    with_hostfed = "hostfed" in also
    with_rtl8 = "rtl8" in also and world == 1
    also = [a for a in also if a not in ("hostfed", "rtl8")]
    cases = [(args.config, case)] + [(a, dict(CASES[a])) for a in also]
    # with several ranks sharing one GPU (gloo rehearsal) keep the footprint small
    def case_bps(i, c):
        f = c.get("format") or (args.format if i == 0 else "u8")
        return 2 if f == "u8" else 4
    need = max(c["channels"] // (max(1, args.share) if i == 0 else 1) * c["blocks"] * 1024 * c["decim"] * case_bps(i, c)
               for i, (_, c) in enumerate(cases))
    hostfed_need = (args.hostfed_channels or HOSTFED["channels"]) * HOSTFED["call_blocks"] * 1024 * HOSTFED["decim"] * 2 * 3
    if args.hostfed_child:    # two calls' worth on the device (the _dev reference of its gate) + one call of scratch
        J.iq_all = torch.empty(hostfed_need, dtype=torch.uint8, device=dev)
        print("HOSTFED_RESULT " + json.dumps(run_hostfed(J, args, args.steps, args.warmup)), flush=True)
        return
    J.iq_all = torch.empty(need, dtype=torch.uint8, device=dev)

    res = []
    for i, (name, c) in enumerate(cases):
        if i and args.cooldown > 0:
            torch.cuda.synchronize()
            time.sleep(args.cooldown)
        res.append(run_case(J, name, c, args, args.steps, args.warmup, headline=(i == 0)))
        if res[-1] is not None:
            res[-1]["config"]["cooldown_before_s"] = args.cooldown if i else 0.0

    if rank == 0:
        head = res[0]
        # what a pure streaming reader gets out of HBM on this very buffer (outside the timed regions): the
        # practical read ceiling next to the 8 TB/s spec figure
        import ctypes as C
        gbs = C.c_double(0)
        nbytes = min(J.iq_all.numel(), 1 << 34)
        if J.L.acg_probe_read_dev(J.iq_all.data_ptr(), nbytes, 3, C.byref(gbs)) == 0:
            for r in res:
                r["roofline"]["pure_reader_GBs_measured_this_run"] = round(gbs.value, 1)
                r["roofline"]["frac_of_pure_reader"] = round(r["roofline"]["achieved"] / gbs.value, 4)
        rtl8 = None
        if with_rtl8:
            try:
                rtl8 = run_rtl8(J, args)
            except SystemExit:
                raise
            except Exception as ex:                   # (a missing demo binary or reference build must not take the line with it)
                rtl8 = {"error": _short(repr(ex), 300)}
        hostfed = None
        if with_hostfed:
            # the hostfed case in a child process with a timeout, after this process has let go of its input buffer
            del J.iq_all
            torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--hostfed-child", "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--sustain", str(args.sustain), "--check-channels", str(args.check_channels)]
            if args.hostfed_channels:
                cmd += ["--hostfed-channels", str(args.hostfed_channels)]
            try:
                r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                got_ = [l for l in r_.stdout.splitlines() if l.startswith("HOSTFED_RESULT ")]
                hostfed = json.loads(got_[-1][len("HOSTFED_RESULT "):]) if (r_.returncode == 0 and got_) else \
                    {"error": "child exited with %d: %s" % (r_.returncode, (r_.stderr or r_.stdout)[-300:])}
            except subprocess.TimeoutExpired:
                hostfed = {"error": "hostfed child did not finish within 240 s"}
        # roofline.traffic of the headline's launch shape, measured here and now (the committed passes stay as the fall-back)
        if world == 1 and not args.no_live_traffic and not overridden and args.format == "u8" and args.share == 1:
            if "iq_all" in J.__dict__:
                del J.iq_all
                torch.cuda.empty_cache()
            rf_ = head["roofline"]
            lt, src = live_traffic(args.config, rf_["kernel"], head["config"]["channels_per_gpu"],
                                   head["config"]["blocks_per_pass"] / max(1, rf_.get("launches_per_pass") or 1))
            rf_["traffic_committed_passes"] = rf_.get("traffic")
            if lt is not None:
                rf_["traffic"], rf_["traffic_source"], rf_["traffic_live"] = lt, src, True
            else:
                rf_["traffic_live"], rf_["traffic_live_note"] = False, src
        out = {
            "metric": "acars_channels_x_input_msps",
            "value": head["value"],
            "unit": "channel*Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
        }
        for k in ("data", "config", "roofline", "whole_job_frac_of_hbm", "whole_job_GBs_per_gpu", "time_dominant_kernel",
                  "timed_region_s", "sustain", "burst", "kernels", "parity", "per_gpu", "valu", "ab_same_process", "placement_trials"):
            if k in head:
                out[k] = head[k]
        if out["time_dominant_kernel"] != out["roofline"]["kernel"]:
            out["roofline"]["note_dominance"] = ("the roofline kernel is the HBM-bound stage; in this case the step time is set by %s (a per-channel "
                                                 "serial recurrence, latency-bound), see whole_job_frac_of_hbm" % out["time_dominant_kernel"])
        if len(res) > 1:
            out["also"] = {name: {k: r[k] for k in ("value", "ms_per_step", "timed_region_s", "sustain", "burst", "whole_job_frac_of_hbm", "whole_job_GBs_per_gpu",
                                                    "time_dominant_kernel", "roofline", "kernels", "parity", "config", "data", "hostfed") if k in r}
                           for (name, _), r in zip(cases[1:], res[1:])}
            for name, r in zip([n for n, _ in cases[1:]], res[1:]):
                if "per_gpu" in r:
                    out["also"][name]["per_gpu"] = r["per_gpu"]
        if hostfed is not None:
            out.setdefault("also", {})["hostfed"] = hostfed
        if rtl8 is not None:
            out.setdefault("also", {})["rtl8"] = rtl8
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(case["decim"])
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        out["multi_gpu"] = MULTI_GPU_NOTE
        # 1. the full detail: an EARLIER stdout line (not JSON-looking: it does not start with "{") and a file
        detail = json.dumps(out)
        print("# bench_detail: " + detail, flush=True)
        paths = [args.detail_file] if args.detail_file else [os.path.join(ROOT, "bench_detail.json")] + (
            [os.path.join(ROOT, "gpurun_out", "bench_detail.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])
        for pth in paths:
            try:
                with open(pth, "w") as f:
                    f.write(detail + "\n")
            except OSError:
                pass
        # 2. the result: ONE compact line, LAST on stdout -- printed below, after the process group has been torn down and C
        # stdio has been flushed (RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would
        # otherwise land behind this line at exit)
        final_line = json.dumps(compact_line(out), separators=(",", ":"))
        assert len(final_line) < 4096, len(final_line)
    if J.coll is not None:
        dist.barrier(device_ids=[local]) if J.backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
