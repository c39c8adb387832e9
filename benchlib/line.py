"""The driver-facing result line (< 4 KB) out of the full detail."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter


MULTI_GPU_NOTE = ("no N>1 run has been measured by the builder (1-GPU boxes only): --gpus N shards channel c to rank c mod N "
                  "(weak scaling, no data-path collective; RCCL carries the 32 B/channel config-table broadcast, barriers and reductions)")



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
    line["data"] = _short(full.get("data", "synthetic"), 110)
    line["config"] = {"workload": _short(cfg.get("workload", ""), 120), "case": cfg.get("case"), "channels_per_gpu": cfg.get("channels_per_gpu"),
                      "decim": cfg.get("decim"), "ntaps": cfg.get("ntaps"), "callbacks_per_call": cfg.get("callbacks_per_call"), "collect_lag": cfg.get("collect_lag"),
                      "passes_per_step": cfg.get("passes_per_step"), "input_format": cfg.get("input_format"),
                      "delivered": _short(cfg.get("delivered", ""), 16)}
    if cfg.get("placement"):
        line["config"]["placement_ms_per_call"] = cfg["placement"].get("ms_per_call")
    line["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_src", "bytes_per_launch", "avg_launch_ms",
                                                "launches_per_step", "pure_reader_GBs_measured_this_run")}
    rm = full.get("roofline_msk")
    if rm:                       # the stage that sets the step of the <= 2048-channel cases: issue-bound, not HBM-bound
        line["roofline_msk"] = {k: rm.get(k) for k in ("bound", "us_per_bit", "cycles_per_bit_per_wave", "instr_per_bit", "frac")}
    for k in ("whole_job_frac_of_hbm", "time_dominant_kernel", "timed_region_s", "per_gpu"):
        if k in full:
            line[k] = full[k]
    line["parity"] = parity_short(full.get("parity"))
    tsrc = {(rf.get("traffic_src") or "none"): [cfg.get("case") or "headline"]}
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
            par = a.get("parity") or {}
            su = a.get("sustain") or {}
            tr, bl = ar.get("traffic"), ar.get("bytes_per_launch")
            # per case: value (channel*Msps), ch(annels) / M (rtlMult) / fmt, the down-converter's roofline fraction and the whole job's
            # fraction of 8 TB/s, traffic_x = PMC HBM bytes / algorithmic bytes of the launch (traffic_src: "live" or the committed pass),
            # the gate's verdicts; s = seconds the value was sustained over, b5 = the rate of its first 5 s, sclk / pw = shader clock (MHz)
            # and power (W) at start / middle / end
            e = {"value": (int(round(a["value"])) if a.get("value") is not None else None), "ch": a.get("config", {}).get("channels_per_gpu"),
                 "M": a.get("config", {}).get("decim"), "whole_job_frac": a.get("whole_job_frac_of_hbm"), "roofline_frac": ar.get("frac"),
                 "traffic_x": (round(tr / bl, 4) if (tr and bl) else None),
                 "parity_ok": parity_ok(a.get("parity")), "blocks": par.get("blocks"),
                 "gpu_vs_ref_ofast": (par.get("end_to_end") or {}).get("gpu_vs_ref_ofast")}
            tsrc.setdefault(ar.get("traffic_src") or ("committed" if tr else "none"), []).append(name)
            if e["M"] == 200:
                del e["M"]                      # (rtlMult 200 = 2.5 Msps unless the entry says otherwise)
            if a.get("config", {}).get("input_format") not in (None, "u8"):
                e["fmt"] = a["config"]["input_format"]
            if a.get("config", {}).get("channels_per_stream"):
                e["ch_per_stream"] = a["config"]["channels_per_stream"]
                ve = ar.get("valu_equivalent") or a.get("valu") or {}
                e["valu_equiv_frac"] = ve.get("frac")
                e["mfma_frac"] = (ar.get("mfma_i8") or {}).get("frac")
                e["kernel"] = _short((ar.get("kernel") or ""), 24)
            if su.get("burst5s"):
                t3 = su.get("telemetry_start_mid_end") or []
                e["s"] = round(a.get("timed_region_s") or 0.0, 1)
                e["b5"] = int(round(su["burst5s"]))
                e["sclk"] = [(t or {}).get("sclk") for t in t3]
                e["pw"] = [(int(round(t["power_w"])) if (t or {}).get("power_w") is not None else None) for t in t3]
            if a.get("config", {}).get("placement"):
                e["placement_ms_per_call"] = a["config"]["placement"].get("ms_per_call")
            if "hostfed" in a:
                e["hostfed"] = {k: a["hostfed"].get(k) for k in ("input_GBs", "frac_of_h2d", "realtime")}
            if "per_gpu" in a:
                e["per_gpu"] = a["per_gpu"]
            line["also"][name] = e
    # where every case's roofline.traffic comes from: "live" = PMC passes inside this invocation, "rNN" = the committed passes of that
    # round (profiles/pmc_traffic.json), "none" = no PMC pass for this launch shape
    line["traffic_src"] = tsrc
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample", ""), 60), "all_cores": cb.get("all_cores"), "gpu_over_cpu": cb.get("gpu_over_cpu")}
    line["multi_gpu"] = _short(full.get("multi_gpu", ""), 120)
    line["detail"] = "bench_detail.json"
    # the budget is enforced, not hoped for: optional keys go, least important first, until the line fits
    size = lambda: len(json.dumps(line, separators=(",", ":")))
    also = line.get("also", {})
    trims = ([lambda a=a: a.pop("placement_ms_per_call", None) for a in also.values()] +
             [lambda a=a: a.pop("kernel", None) for a in also.values()] +
             [lambda a=a: a.pop("blocks", None) for a in also.values()] +
             [lambda a=a: a.__setitem__("per_gpu", [int(round(x)) for x in a["per_gpu"]]) if "per_gpu" in a else None for a in also.values()] +
             [lambda: line.__setitem__("data", _short(line["data"], 60)),
              lambda: line["config"].__setitem__("workload", _short(line["config"]["workload"], 160)),
              lambda: line.__setitem__("multi_gpu", _short(line["multi_gpu"], 60))] +
             [lambda a=a: a.pop("pw", None) for a in also.values()] +
             [lambda a=a: a.pop("sclk", None) for a in also.values()] +
             [lambda a=a: a.pop("per_gpu", None) for a in also.values()])
    for t in trims:
        if size() <= 4000:
            break
        t()
    return line
