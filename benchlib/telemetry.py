"""Shader clock / power / temperature of the card a rank runs on, from sysfs (null where the box does not expose them)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")          # the entry point the child processes of a run re-enter


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
