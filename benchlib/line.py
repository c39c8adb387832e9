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
