"""Turns the PMC summaries of one round (profiles/<tag>_<case>_fetch.txt / _write.txt, made by profiles/summarize_rocpd.py from
separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over `bench.py --config <case> --also none`) and the bench
line of the same launch shape into entries of profiles/pmc_traffic.json:

    python profiles/pmc_to_json.py <tag> <case>=<bench line json> [...]

traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 bytes per launch (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE counts
wide coalesced reads at half their bytes; WRITE_SIZE was calibrated exact on a known-size fill, see _about in the json)."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
path = os.path.join(HERE, "pmc_traffic.json")
doc = json.load(open(path))


def counter(fn, kernel_prefix, name):
    for l in open(fn):
        if name in l and kernel_prefix in l:
            m = re.search(name + r"\s+(\d+)\s+([\d.]+)", l)
            return int(m.group(1)), float(m.group(2))
    return None, None


for spec in sys.argv[2:]:
    case, line = spec.split("=", 1)
    d = json.loads([x for x in open(line) if x.startswith("{")][-1])
    det = None
    for x in open(line):
        if x.startswith("# bench_detail: "):
            det = json.loads(x[len("# bench_detail: "):])
    rf = (det or d)["roofline"]
    cfg = (det or d)["config"]
    kern = rf["kernel"]
    prefix = kern.split("<")[0] + "<" + kern.split("<")[1][:12] if "<" in kern else kern
    n1, fetch = counter(os.path.join(HERE, "%s_%s_fetch.txt" % (tag, case)), kern.split(">")[0][:40], "FETCH_SIZE")
    n2, write = counter(os.path.join(HERE, "%s_%s_write.txt" % (tag, case)), kern.split(">")[0][:40], "WRITE_SIZE")
    if fetch is None or write is None:
        print("no counters for", case, kern)
        continue
    lpp = rf.get("launches_per_pass") or rf["launches_per_step"]
    e = {"kernel": kern, "channels": cfg["channels_per_gpu"], "decim": cfg["decim"], "ntaps": cfg["ntaps"],
         "blocks_per_launch": cfg["blocks_per_pass"] / lpp, "fetch_size_kb": fetch, "write_size_kb": write,
         "traffic_bytes": int(round(2 * fetch * 1024 + write * 1024)), "algorithmic_bytes": rf["bytes_per_launch"],
         "launches_counted": [n1, n2],
         "source": "profiles/%s_%s_fetch.txt, profiles/%s_%s_write.txt (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, over "
                   "`bench.py --config %s --also none --sustain 0`)" % (tag, case, tag, case, case)}
    e["ratio"] = round(e["traffic_bytes"] / e["algorithmic_bytes"], 4)
    doc["entries"] = [x for x in doc["entries"] if not (x["kernel"] == e["kernel"] and (x["channels"], x["decim"], x["ntaps"], x["blocks_per_launch"]) ==
                                                        (e["channels"], e["decim"], e["ntaps"], e["blocks_per_launch"]))]
    doc["entries"].insert(0, e)
    print(case, kern, "traffic / algorithmic =", e["ratio"])
json.dump(doc, open(path, "w"), indent=1)
