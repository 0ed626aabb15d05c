#!/usr/bin/env python3
"""Background sampler of socket power / gfx clocks / activity through amdsmi (ROCm's SMI library), one JSON line per sample:

    python tools/smi_sampler.py OUT.jsonl [period_s=0.02]      (runs until killed; the first line is the full metrics dict)

Used by tools/power_clock_lab.py: the load runs in another process and records wall-clock windows (time.time()); the table joins
them with these samples.  Falls back to the hwmon files of the first amdgpu card when amdsmi is unusable."""
import glob
import json
import sys
import time


def hwmon_reader():
    base = None
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if glob.glob(d + "/power1_*"):
            base = d
            break
    if base is None:
        return None

    def rd(name):
        try:
            return int(open(f"{base}/{name}").read().strip())
        except Exception:
            return None

    def sample():
        p = rd("power1_average")
        if p is None:
            p = rd("power1_input")
        return {"power_w": None if p is None else p / 1e6, "gfxclk_mhz": None if rd("freq1_input") is None else rd("freq1_input") / 1e6, "src": "hwmon"}

    return sample


def amdsmi_reader(first_out):
    import amdsmi

    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]

    def clean(v):
        if isinstance(v, (list, tuple)):
            return [clean(x) for x in v]
        if isinstance(v, dict):
            return {k: clean(x) for k, x in v.items()}
        if isinstance(v, (int, float, str)) or v is None:
            return v
        return str(v)

    try:
        first_out["gpu_metrics"] = clean(amdsmi.amdsmi_get_gpu_metrics_info(h))
    except Exception as e:  # noqa: BLE001
        first_out["gpu_metrics_error"] = repr(e)
    try:
        first_out["power_info"] = clean(amdsmi.amdsmi_get_power_info(h))
    except Exception as e:  # noqa: BLE001
        first_out["power_info_error"] = repr(e)
    try:
        first_out["clock_info"] = clean(amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
    except Exception as e:  # noqa: BLE001
        first_out["clock_info_error"] = repr(e)
    try:
        first_out["power_cap"] = clean(amdsmi.amdsmi_get_power_cap_info(h))
    except Exception as e:  # noqa: BLE001
        first_out["power_cap_error"] = repr(e)

    def num(x):
        return x if isinstance(x, (int, float)) and x < 60000 else None

    def sample():
        out = {"src": "amdsmi"}
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            out["power_w"] = num(m.get("current_socket_power"))
            if out["power_w"] is None:
                out["power_w"] = num(m.get("average_socket_power"))
            clks = m.get("current_gfxclks")
            if isinstance(clks, (list, tuple)):
                v = [c for c in clks if num(c)]
                out["gfxclks_mhz"] = v[:8]
                out["gfxclk_mhz"] = sum(v[:8]) / max(1, len(v[:8])) if v else None
            else:
                out["gfxclk_mhz"] = num(m.get("current_gfxclk"))
            out["uclk_mhz"] = num(m.get("current_uclk"))
            out["socclk_mhz"] = num(m.get("current_socclk"))
            out["gfx_activity"] = num(m.get("average_gfx_activity"))
            out["umc_activity"] = num(m.get("average_umc_activity"))
            out["temp_hotspot"] = num(m.get("temperature_hotspot"))
            out["throttle_status"] = clean(m.get("throttle_status"))
            out["energy_acc"] = clean(m.get("energy_accumulator"))
            out["indep_throttle"] = clean(m.get("indep_throttle_status"))
        except Exception as e:  # noqa: BLE001
            out["metrics_error"] = repr(e)
            try:
                p = amdsmi.amdsmi_get_power_info(h)
                out["power_w"] = num(p.get("current_socket_power")) or num(p.get("average_socket_power"))
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                out["gfxclk_mhz"] = num(c.get("clk"))
            except Exception as e2:  # noqa: BLE001
                out["error"] = repr(e2)
        return out

    return sample


def main():
    path = sys.argv[1]
    period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
    first = {"t": time.time(), "first": True}
    try:
        sample = amdsmi_reader(first)
    except Exception as e:  # noqa: BLE001
        first["amdsmi_error"] = repr(e)
        sample = hwmon_reader()
    with open(path, "w", buffering=1) as f:
        f.write(json.dumps(first) + "\n")
        if sample is None:
            f.write(json.dumps({"t": time.time(), "error": "no power source"}) + "\n")
            return
        while True:
            t = time.time()
            s = sample()
            s["t"] = t
            s["dt_read"] = time.time() - t
            f.write(json.dumps(s) + "\n")
            time.sleep(max(0.0, period - (time.time() - t)))


if __name__ == "__main__":
    main()
