"""Board power / shader clock of ONE GPU from its hwmon node, sampled by a thread of the calling process (round 6).

What an ordinary user can read on the pool's boxes (tools/probe_telemetry.sh): ``/sys/class/drm/card*/device/hwmon/hwmon*/``
``power1_input`` (socket power, microwatts), ``power1_cap`` (the board limit: 1 400 W), ``freq1_input`` (sclk, Hz).  All eight cards of
the host are visible there; the one this process computes on is found by its PCI bus id.  Reading sysfs initialises nothing on the GPU
and starts no process, so the sampler may run under rocprofv3 and beside a HIP-graph capture.

    with PowerSampler(pci_bus_id_of(0)) as ps:
        ... GPU work ...
    ps.summary()  ->  {"power_w_mean", "power_w_max", "sclk_mhz_mean", "sclk_mhz_min", "power_cap_w", "samples", ...}
"""
import glob
import os
import threading
import time


def pci_bus_id_of(index: int = 0):
    """'0000:75:00.0' of HIP device `index` (torch's device properties), or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        dom = getattr(p, "pci_domain_id", 0)
        return f"{dom:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


def hwmon_dir(pci: str = None):
    """The hwmon directory of the card at PCI address `pci`; without one: the only card, or None when several are visible."""
    cands = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if not os.path.exists(os.path.join(h, "power1_input")):
            continue
        dev = os.path.realpath(os.path.join(h, "..", ".."))
        cands.append((os.path.basename(dev), h))
    if pci:
        for addr, h in cands:
            if addr.lower() == pci.lower():
                return h
        return None
    return cands[0][1] if len(cands) == 1 else None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


class PowerSampler:
    def __init__(self, pci: str = None, period_s: float = 0.01):
        self.dir = hwmon_dir(pci)
        self.pci = pci
        self.period = period_s
        self.t, self.w, self.mhz = [], [], []
        self._stop = threading.Event()
        self._th = None
        self.cap_w = None
        if self.dir:
            c = _read_int(os.path.join(self.dir, "power1_cap"))
            self.cap_w = c / 1e6 if c else None

    @property
    def available(self):
        return self.dir is not None

    def _run(self):
        pw, fq = os.path.join(self.dir, "power1_input"), os.path.join(self.dir, "freq1_input")
        while not self._stop.is_set():
            p, f = _read_int(pw), _read_int(fq)
            if p is not None:
                self.t.append(time.perf_counter())
                self.w.append(p / 1e6)
                self.mhz.append(f / 1e6 if f else float("nan"))
            self._stop.wait(self.period)

    def start(self):
        if self.dir and self._th is None:
            self._stop.clear()
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def stop(self):
        if self._th is not None:
            self._stop.set()
            self._th.join()
            self._th = None
        return self

    __enter__ = start

    def __exit__(self, *a):
        self.stop()

    def mark(self):
        """number of samples so far: summary(lo=mark_a, hi=mark_b) describes the samples taken between two marks"""
        return len(self.w)

    def summary(self, lo: int = 0, hi: int = None, skip_s: float = 0.0):
        """mean / max over samples [lo, hi); skip_s drops the first seconds of the window (the ramp: the SMU's reading is a moving average)"""
        if not self.dir:
            return {"available": False, "why": f"no hwmon node for PCI device {self.pci}"}
        hi = len(self.w) if hi is None else hi
        idx = [i for i in range(lo, hi) if self.t[i] - self.t[lo] >= skip_s] if hi > lo else []
        if not idx:
            return {"available": True, "samples": 0, "power_cap_w": self.cap_w}
        w = [self.w[i] for i in idx]
        m = [self.mhz[i] for i in idx if self.mhz[i] == self.mhz[i]]
        return {"available": True, "samples": len(w), "power_w_mean": sum(w) / len(w), "power_w_max": max(w), "power_w_min": min(w),
                "sclk_mhz_mean": (sum(m) / len(m)) if m else None, "sclk_mhz_min": min(m) if m else None, "sclk_mhz_max": max(m) if m else None,
                "power_cap_w": self.cap_w, "seconds": self.t[idx[-1]] - self.t[idx[0]], "source": os.path.join(self.dir, "{power1_input,freq1_input}")}
