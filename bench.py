#!/usr/bin/env python
"""bench.py -- whole-job throughput of the hot path (rtl.c down-converter + msk.c demodulator +
framing FSM) on N GPUs of one node, with the roofline of the dominant kernel and a CPU baseline.

A "step" is one pass of the hot path over one batch: `--channels` independent 2.5 Msps u8 I/Q
streams per GPU (one stream per channel, BASELINE.json configs[2]: 1024 channels, rtlMult=200),
`--blocks` reference callbacks (1024 outputs each = 81.92 ms of signal) per channel, inputs already
resident in HBM.  Every step ends with the decoded blocks drained to the host.

Multi-GPU: channels are independent, so each rank owns its own channels, input and
state; there is no data-path collective.  RCCL carries only the barrier and the reduction of the
timing / counts (scaling: weak, per-GPU work fixed).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


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
    fc = 131725000 + 25000
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
        cmd = [sys.executable, me, "--cpu-child", variant, str(M), "4", str(seconds)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0 and r.stdout.strip():
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out = dict(value=round(d["value"], 2), unit="channel*Msamples/s", cores=1, kind="reference",
                       sample="unmodified reference rtl.c in_callback + msk.c + acars.c (%s), 1 channel per stream, "
                              "rtlMult=%d, %d callbacks of 1024 outputs in %.1f s on one host core (the reference is single-threaded)"
                              % (label, M, d["blocks"], d["seconds"]))
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


PRESETS = {
    # BASELINE.json configs[2]: 1 GPU, 1024 channels, synthetic 2.5 Msps IQ, FIR decimate + MSK demod throughput
    "throughput": dict(channels=1024, decim=200, ntaps=200, blocks=8),
    # BASELINE.json configs[4]: 1 GPU stress, 192-tap LPF FIR, 2.5 Msps, 4096 channels
    "stress": dict(channels=4096, decim=200, ntaps=192, blocks=4),
    # BASELINE.json configs[3] per-GPU share: 16384 channels over 8 GPUs
    "shard2048": dict(channels=2048, decim=200, ntaps=200, blocks=8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(PRESETS), default="throughput")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (overrides the preset)")
    ap.add_argument("--decim", type=int, default=None, help="rtlMult: 200 = 2.5 Msps")
    ap.add_argument("--ntaps", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None, help="1024-output callbacks per channel per step")
    ap.add_argument("--check-channels", type=int, default=24, help="channels of rank 0 verified against the oracle")
    ap.add_argument("--random-bytes", action="store_true", help="uniform random input bytes instead of ACARS traffic")
    ap.add_argument("--format", choices=["u8", "cs16", "split16", "f32"], default="u8",
                    help="input sample format: u8 = rtl.c (headline); cs16 = soapy.c, split16 = sdrplay.c, f32 = air.c (SURVEY 8f.2)")
    ap.add_argument("--share", type=int, default=1,
                    help="channels per input stream (rtl.c's own shape: one dongle feeds up to 16 channels); >1 = shared-stream "
                         "mode, VALU-bound, reported separately and never as the roofline figure (SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-child", nargs=4, default=None)
    args = ap.parse_args()
    if args.cpu_child:
        v, M, b, s = args.cpu_child
        cpu_baseline_child(v, int(M), int(b), float(s))
        return
    pre = PRESETS[args.config]
    nch = args.channels or pre["channels"]
    M = args.decim or pre["decim"]
    ntaps = args.ntaps or (pre["ntaps"] if args.decim is None else M)
    nblk = args.blocks or pre["blocks"]

    import numpy as np
    import torch
    import torch.distributed as dist
    from acarsdec_amd import decoder as D, synth as S, _capi as K, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local = local % torch.cuda.device_count()       # (a rehearsal of the N > 1 path on a 1-GPU box maps all ranks to GPU 0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("ACG_BENCH_BACKEND", "nccl")    # "gloo": rehearsal without RCCL (several ranks on one GPU)
    cdev = dev if backend == "nccl" else None                # where the few collective tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    L = K.load()

    # ---- per-channel configuration: made on rank 0 for ALL channels of the job, scattered over RCCL
    # (the only data that ever crosses xGMI: ~32 B per channel; inputs are generated where they are used)
    nch_total = nch * world
    NPOOL = 64
    cfg_rows = None
    if rank == 0:
        r0 = np.random.default_rng(0xACA25)
        off = r0.integers(-48, 49, size=nch_total) * 25000.0          # multiples of 12.5 kHz within +-1.2 MHz
        off[np.abs(off) < 25000] = 50000.0                             # >= 25 kHz from DC like chooseFc enforces
        cfg_rows = np.stack([off, r0.uniform(0, 2 * np.pi, nch_total), r0.integers(0, NPOOL, nch_total).astype(np.float64),
                             np.arange(nch_total, dtype=np.float64)], axis=1)
    mine = shard.scatter_channel_config(cfg_rows, world, rank, dist if world > 1 else None, device=cdev)
    own = shard.owned_channels(nch_total, rank, world)
    assert mine.shape[0] == nch and np.array_equal(mine[:, 3].astype(np.int64), own)
    offs, phases, pool_idx = mine[:, 0], mine[:, 1], mine[:, 2].astype(np.int32)

    fc = 131000000
    taps = np.zeros((nch, ntaps, 2), dtype=np.float32)
    win = np.ones(ntaps) if ntaps == M else np.hamming(ntaps) / np.hamming(ntaps).mean() * (M / ntaps)
    tap_cache = {}
    for c in range(nch):
        o = int(offs[c])
        if o not in tap_cache:
            # the front end's own tap builder (rtl.c:283-286 / soapy.c:163-166 / air.c:278-285)
            base = (D.rtl_taps(fc + o, fc, M) if args.format == "u8" else
                    D.airspy_taps(fc - o, fc, M * 12500) if args.format == "f32" else D.soapy_taps(fc + o, fc, M))
            tap_cache[o] = (base[:ntaps] * win[:, None]).astype(np.float32)
        taps[c] = tap_cache[o]

    # ---- inputs, resident in HBM: distinct bytes per channel, working set >> 256 MiB Infinity Cache
    fmt = {"u8": 0, "cs16": K.FMT_CS16, "split16": K.FMT_S16_SPLIT, "f32": K.FMT_F32_REAL}[args.format]
    bps = 2 if fmt == 0 else 4
    row = nblk * 1024 * M * bps
    share = max(1, args.share)
    if share > 1:
        assert fmt == 0 and nch % share == 0, "--share needs the u8 format and a channel count divisible by it"
        args.random_bytes = True
    nstreams = nch // share
    iq = torch.empty((nstreams, row), dtype=torch.uint8, device=dev)
    if fmt == K.FMT_F32_REAL:
        iq.view(torch.float32).normal_(0.0, 0.1)
        data_desc = "gaussian float32 samples (format throughput run; parity of this format is covered by tests/)"
    elif fmt != 0:
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nch, row, 0xACA25 + rank, None) == 0
        iq.view(torch.int16).bitwise_and_(0x0FFF)
        data_desc = "uniform random int16 samples (format throughput run; parity of this format is covered by tests/)"
    elif args.random_bytes:
        assert L.acg_fill_random_u8_dev(iq.data_ptr(), row, nstreams, row, 0xACA25 + rank, None) == 0
        data_desc = "uniform random bytes"
    else:
        # a pool of ACARS/MSK audio tracks (SURVEY App. C.2 modulator), every channel = one track on its
        # own carrier offset / phase / noise realisation, up-converted on the device
        prng = np.random.default_rng(0xACA25)
        pool = np.zeros((NPOOL, nblk * 1024), dtype=np.float32)
        for i in range(NPOOL):
            a, _ = S.channel_audio(prng, nblk * 1024, gap=(1500, 5000), text_len=(20, 160))
            pool[i] = 0.5 * (1.0 + 0.5 * a)
        d_pool = torch.from_numpy(pool).to(dev)
        d_idx = torch.from_numpy(pool_idx).to(dev)
        d_off = torch.from_numpy(offs.astype(np.float32)).to(dev)
        d_ph = torch.from_numpy(phases.astype(np.float32)).to(dev)
        rc = L.acg_synth_iq_u8_dev(iq.data_ptr(), row, nch, nblk * 1024, M, d_pool.data_ptr(), pool.shape[1], d_idx.data_ptr(),
                                   d_off.data_ptr(), d_ph.data_ptr(), 0.25, 0.05, 0xACA25 + rank, None)
        assert rc == 0, rc
        data_desc = "ACARS/MSK traffic on every channel (%d-track pool, AM depth 0.5, AWGN sigma 0.05, device-side up-converter)" % NPOOL
    torch.cuda.synchronize()

    dec = D.Decoder(nch, decim=M, ntaps=ntaps, nstreams=nstreams, max_blocks=nblk, device=local, bitlog=True, timing=True)
    dec.set_taps(taps)
    if share > 1:
        dec.set_channel_streams(np.arange(nch) // share)
    stream = torch.cuda.current_stream().cuda_stream

    maxfr = max(8192, 8 * nch)

    def step(lag=1):
        """one pass of the hot path; decoded blocks are delivered to the host one call behind
        (streaming double buffering: the newest call keeps the GPU busy while the host collects)"""
        if fmt == 0:
            dec.in_callback(iq, nblocks=nblk, pitch=row, stream=stream)
        else:
            dec.process_samples(fmt, iq, nblk, pitch=row, plane=row // 2 if fmt == K.FMT_S16_SPLIT else 0, stream=stream)
        return dec.collect_frames_raw(lag, maxfr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    # ---- correctness gate on the first pass (state starts from reset): a subset of rank 0's channels
    # goes through the CPU oracle on the very bytes the GPU consumed
    n_first, fbuf = step(lag=0)
    first = [K.Frame.from_buffer_copy(fbuf[i]) for i in range(n_first)]
    parity = None
    if rank == 0 and fmt == 0:
        from oracle import oracle as O
        ncheck = min(args.check_channels, nch)
        got = {}
        for f in first:
            got.setdefault(int(f.chn), []).append(D.frame_tuple(f))
        ok, nblocks = True, 0
        host_rows = iq[:(ncheck + share - 1) // share].cpu().numpy()
        for c in range(ncheck):
            ch = O.Channel(c)
            ch.demod(O.fir_u8(host_rows[c // share], M, taps[c], ntaps=ntaps))
            want = [O.frame_tuple(f) for f in ch.frames]
            nblocks += len(want)
            ok &= got.get(c, []) == want
        parity = dict(channels_checked=ncheck, blocks=nblocks, bit_exact=bool(ok), blocks_first_pass_all_channels=len(first))
        if not ok:
            raise SystemExit("bench: GPU blocks differ from the oracle: %r" % parity)

    for _ in range(args.warmup):
        step()
    dec.drain_frames_raw(maxfr)       # flush: the timed region starts with empty queues
    warm = dec.timing()               # event sums of warm-up: the demodulator's launches are timed here only --
    dec.set_timing(2)                 # in the timed region only the down-converter (roofline) is bracketed,
                                      # event records on the demodulator stream sit on its serial launch chain
    barrier()
    t0 = time.perf_counter()
    nfr = 0
    for _ in range(args.steps):
        nfr += step()[0]
    nfr += dec.drain_frames_raw(maxfr)[0]      # the last call's blocks: all K steps fully delivered inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    tim = dec.timing()
    dt, nfr_total = shard.reduce_timing(dt, nfr, world, dist if world > 1 else None, cdev)

    # what a plain vendor read-reduction gets out of HBM on this very buffer (outside the timed region):
    # the practical read ceiling next to the 8 TB/s spec figure
    probe_gbs = None
    if rank == 0:
        v64 = iq.view(torch.int64)
        v64.sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            v64.sum()
        e1.record()
        torch.cuda.synchronize()
        probe_gbs = iq.numel() * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    if rank == 0:
        samples_per_step = nch * nblk * 1024 * M                    # complex input samples per GPU per step
        value = world * samples_per_step * args.steps / dt / 1e6    # channel * Msamples/s
        # algorithmic bytes of one FIR launch: 2 B per input sample per channel read,
        # 4 B per 12.5 kHz output written, taps (8 B each) read once per launch.  A step is split
        # into `lps` pipelined FIR launches (chunks of the step's callbacks).
        lps = max(1, round(tim["fir_launches"] / args.steps))
        # (shared-stream mode: each stream's bytes count once)
        fir_bytes = (nstreams * (nblk / lps) * 1024 * bps * M + nch * (nblk / lps) * 1024 * 4) + nch * ntaps * 8
        fir_avg_ms = tim["fir_ms"] / max(1, tim["fir_launches"])
        achieved = fir_bytes / (fir_avg_ms * 1e-3) / 1e9
        # HBM traffic of this launch shape from the committed PMC passes (rocprofv3 cannot run inside the
        # timed process; see profiles/pmc_traffic.json for the counters and the gfx950 correction)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                for e in json.load(f)["entries"]:
                    if fmt == 0 and share == 1 and (e["channels"], e["decim"], e["ntaps"]) == (nch, M, ntaps) \
                            and abs(e["blocks_per_launch"] - nblk / lps) < 1e-9 and e["kernel"].startswith("fir_u8_persist"):
                        traffic = e["traffic_bytes"]
        except Exception:
            traffic = None
        out = {
            "metric": "acars_channels_x_input_msps",
            "value": round(value, 1),
            "unit": "channel*Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: " + data_desc,
            "config": {"workload": ("BASELINE configs[%s]: %d channels/GPU x %.1f Msps FORMAT_TAG input, one stream per channel, rtlMult=%d, ntaps=%d, "
                                    "%d callbacks (%.3f s of signal) per step; FIR decimate + MSK demod + framing, blocks delivered to the host (one call behind)"
                                    % ({"throughput": "2", "stress": "4", "shard2048": "3"}[args.config], nch, 12500 * M / 1e6, M, ntaps, nblk, nblk * 0.08192)).replace(
                                        "FORMAT_TAG", {"u8": "u8 IQ", "cs16": "CS16 IQ", "split16": "split int16 I/Q", "f32": "real f32"}[args.format]),
                       "input_format": args.format, "channels_per_gpu": nch, "decim": M, "ntaps": ntaps, "blocks_per_step": nblk,
                       "realtime_channels_equiv": int(value / (12500 * M / 1e6)),
                       "arithmetic": "u8 in, f32 down-converter and matched filter, f64 VCO/PLL/normalisation (as the reference)",
                       "preset": args.config, "channels_total": nch_total, "blocks_decoded_timed": int(nfr_total)},
            "roofline": {"bound": "hbm", "kernel": ("fir_u8_shared_kernel" if share > 1 else "fir_u8_persist_kernel") if fmt == 0 else "fir_fmt_kernel<%s>" % args.format, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": int(fir_bytes), "avg_launch_ms": round(fir_avg_ms, 4), "launches_per_step": lps,
                         "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4),
                         "read_probe_GBs_torch_sum_same_buffer": round(probe_gbs, 1), "frac_of_read_probe": round(achieved / probe_gbs, 4),
                         "pure_nt_reader_GBs_profiles_probe": 7050.0, "frac_of_pure_nt_reader": round(achieved / 7050.0, 4)},
            "kernels": {"fir_ms_per_step": round(tim["fir_ms"] / args.steps, 4), "msk_ms_per_step": round(warm["msk_ms"] / (args.warmup + 1), 4),
                        "note": "per-step sums of event-timed launches; down-converter chunks overlap the demodulator chunks of the previous chunk; the demodulator figure is taken during warm-up (its events are off in the timed region)"},
            "parity": parity,
        }
        if share > 1:
            out["config"]["channels_per_stream"] = share
            out["roofline"]["note"] = ("shared-stream mode: %d channels reuse each stream's bytes, the down-converter is VALU-bound "
                                       "(8*K flop per 2 B); achieved counts each stream once and is NOT the HBM roofline figure" % share)
            # VALU lane-ops of the shared-stream kernel: per 8 complex samples 24 shared conversion ops + 16 packed
            # FMAs per channel of the group (groups of <= 8); peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
            keff = min(share, 8)
            ops = nch * nblk * 1024 * M * (2.0 + 3.0 / keff) * args.steps / (tim["fir_ms"] * 1e-3)
            out["valu"] = {"kernel": "fir_u8_shared_kernel", "lane_ops_per_s": round(ops, 0), "peak": 256 * 4 * 16 * 2.4e9,
                           "frac": round(ops / (256 * 4 * 16 * 2.4e9), 4), "lane_ops_per_channel_sample": round(2.0 + 3.0 / keff, 3)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(M)
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
