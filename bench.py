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

from benchlib.cases import CASES, COPY_CEILING_GBS, HBM_PEAK_GBS, HOSTFED, Job          # noqa: E402,F401  (tests reach CASES through this module)
from benchlib.case import run_case                                                       # noqa: E402
from benchlib.cpu_baseline import cpu_baseline_child, run_cpu_baseline                   # noqa: E402
from benchlib.hostfed import run_hostfed                                                 # noqa: E402
from benchlib.launch import free_port, self_launch                                       # noqa: E402
from benchlib.line import MULTI_GPU_NOTE, _short, compact_line                           # noqa: E402,F401
from benchlib.rtl8 import rtl8_cpu_child, rtl8_freqs, rtl8_oneline, run_rtl8                # noqa: E402,F401
from benchlib.traffic import live_traffic                                                # noqa: E402


def dry_run_main(J, args, dist, local):
    """--dry-run: the launch path of main() below without a device (see benchlib/dryrun.py); same cases, same line"""
    from benchlib.dryrun import run_case_dry
    world, rank = J.world, J.rank
    J.backend, J.local, J.dev, J.cdev, J.dist, J.coll, J.L = "gloo", local, None, None, dist, None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        J.coll = dist
    also = ["shard2048"] if args.also is None else ([] if args.also in ("", "none") else args.also.split(","))
    cases = [(args.config, dict(CASES[args.config]))] + [(a, dict(CASES[a])) for a in also if a != args.config]
    if args.channels:
        cases[0][1]["channels"] = args.channels
    res = [run_case_dry(J, name, c, args, args.steps, args.warmup, headline=(i == 0)) for i, (name, c) in enumerate(cases)]
    final_line = None
    if rank == 0:
        head = res[0]
        out = {"metric": "acars_channels_x_input_msps", "value": head["value"], "unit": "channel*Msamples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "dry_run": True}
        for k in ("data", "config", "roofline", "whole_job_frac_of_hbm", "time_dominant_kernel", "timed_region_s", "sustain", "parity", "per_gpu"):
            if k in head:
                out[k] = head[k]
        if len(res) > 1:
            out["also"] = {name: r for (name, _), r in zip(cases[1:], res[1:])}
        out["multi_gpu"] = MULTI_GPU_NOTE
        print("# bench_detail: " + json.dumps(out), flush=True)
        line = compact_line(out)
        line["dry_run"] = True
        final_line = json.dumps(line, separators=(",", ":"))
        assert len(final_line) < 4096, len(final_line)
    if J.coll is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(final_line, flush=True)


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
    ap.add_argument("--sustain-hbm", type=float, default=20.0,
                    help="seconds the timed region of the cases that saturate HBM (benchlib.cases.SUSTAIN_HBM: wide, stress, cs16, f32) lasts at "
                         "least: at the power cap their rate sinks with the shader clock for tens of seconds, the first 5 s are reported "
                         "beside it as burst5s")
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
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the N-rank launch path without a GPU (benchlib/dryrun.py): real launcher, real scatter / barriers / "
                         "reductions over gloo, real timed region and result line, a stub that sleeps in place of the decoder.  The line "
                         "says \"dry_run\": true and measures nothing")
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
    if args.dry_run:
        return dry_run_main(J, args, dist, local)
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
        also = (["shard2048", "share8", "wide", "stress", "m160", "m192", "cs16", "f32", "split16", "rtl8", "hostfed"] if world == 1
                else ["shard2048"])
        also = [a for a in also if a != args.config]
    with_hostfed = "hostfed" in also
    with_rtl8 = "rtl8" in also and world == 1
    also = [a for a in also if a not in ("hostfed", "rtl8")]
    cases = [(args.config, case)] + [(a, dict(CASES[a])) for a in also]
    # with several ranks sharing one GPU (gloo rehearsal) keep the footprint small
    def case_bps(i, c):
        f = c.get("format") or (args.format if i == 0 else "u8")
        return 2 if f == "u8" else 4
    need = max(c["channels"] // int(c.get("share") or (max(1, args.share) if i == 0 else 1)) * c["blocks"] * 1024 * c["decim"] * case_bps(i, c)
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
                rf_["traffic"], rf_["traffic_source"], rf_["traffic_live"], rf_["traffic_src"] = lt, src, True, "live"
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
                  "timed_region_s", "sustain", "burst", "kernels", "parity", "per_gpu", "valu", "ab_same_process", "placement_trials", "roofline_msk"):
            if k in head:
                out[k] = head[k]
        if out["time_dominant_kernel"] != out["roofline"]["kernel"]:
            out["roofline"]["note_dominance"] = ("the roofline kernel is the HBM-bound stage; in this case the step time is set by %s (a per-channel "
                                                 "serial recurrence, latency-bound), see whole_job_frac_of_hbm" % out["time_dominant_kernel"])
        if len(res) > 1:
            out["also"] = {name: {k: r[k] for k in ("value", "ms_per_step", "timed_region_s", "sustain", "burst", "whole_job_frac_of_hbm", "whole_job_GBs_per_gpu",
                                                    "time_dominant_kernel", "roofline", "roofline_msk", "kernels", "parity", "config", "data", "hostfed") if k in r}
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
