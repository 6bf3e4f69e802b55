"""Power / clock trace of the GPU while a command runs ("commit the smi power / sclk trace beside the run to
prove or kill the power-limited claim").  Samples the amdgpu hwmon / sysfs nodes of card 0 at ~20 Hz (no rocm-smi start-up cost per
sample; falls back to `rocm-smi --showpower --showclocks --json` at ~2 Hz when the nodes are absent) and writes a CSV plus a
summary JSON: per phase (the command may print lines `##PHASE name` on stdout to label what it is doing) mean / max power and the
mean shader clock.

    python tools/smi_trace.py out_prefix -- python tools/smi_phases.py"""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _nodes():
    """hwmon nodes of EVERY amdgpu card visible in sysfs (the container sees all of the host's cards there, but only one through
    HIP): all are sampled, and the summary names the one whose power follows the phases as the active card"""
    cards = {}
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        card = hw.split("/")[4]
        n = {}
        for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)), ("temp_mC", ("temp1_input", "temp2_input")),
                           ("cap_uW", ("power1_cap",))):
            for nm in names:
                p = os.path.join(hw, nm)
                if key not in n and os.path.exists(p):
                    n[key] = p
        if "power_uW" in n:
            cards[card] = n
    return cards


def _read(p):
    try:
        return float(open(p).read().strip())
    except Exception:
        return None


def _smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = next(iter(d.values()))
        pw = next((float(v) for k, v in c.items() if "Power" in k and "W" in k), None)
        sclk = next((float(str(v).strip("()Mhz ")) for k, v in c.items() if k.startswith("sclk clock speed")), None)
        return pw, sclk
    except Exception:
        return None, None


def main():
    prefix = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cards = _nodes()
    rows, phase, stop = [], ["start"], [False]

    def sampler():
        t0 = time.time()
        while not stop[0]:
            if cards:
                for card, nodes in cards.items():
                    pw = _read(nodes["power_uW"])
                    sc = _read(nodes["sclk_Hz"]) if nodes.get("sclk_Hz") else None
                    tm = _read(nodes["temp_mC"]) if nodes.get("temp_mC") else None
                    rows.append((time.time() - t0, phase[0], card, pw / 1e6 if pw else None, sc / 1e6 if sc else None, tm / 1e3 if tm else None))
                time.sleep(0.05)
            else:
                pw, sc = _smi_sample()
                rows.append((time.time() - t0, phase[0], "rocm-smi", pw, sc, None))
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in p.stdout:
        if line.startswith("##PHASE "):
            phase[0] = line.split(None, 1)[1].strip()
        else:
            lines.append(line)
    p.wait()
    stop[0] = True
    th.join(timeout=15)
    with open(prefix + ".csv", "w") as f:
        f.write("t_s,phase,card,power_W,sclk_MHz,temp_C\n")
        for r in rows:
            f.write(",".join("" if x is None else (x if isinstance(x, str) else "%.4g" % x) for x in r) + "\n")
    summ = {"source": "sysfs hwmon of every amdgpu card" if cards else "rocm-smi --json", "samples": len(rows), "cards": {}}
    best, best_swing = None, -1.0
    for card in dict.fromkeys(r[2] for r in rows):
        cs = {"power_cap_W": (_read(cards[card]["cap_uW"]) / 1e6 if cards.get(card, {}).get("cap_uW") else None), "phases": {}}
        for ph in dict.fromkeys(r[1] for r in rows):
            pw = [r[3] for r in rows if r[1] == ph and r[2] == card and r[3] is not None]
            sc = [r[4] for r in rows if r[1] == ph and r[2] == card and r[4] is not None]
            cs["phases"][ph] = {"samples": len(pw), "power_W_mean": sum(pw) / len(pw) if pw else None, "power_W_max": max(pw) if pw else None,
                                "sclk_MHz_mean": sum(sc) / len(sc) if sc else None, "sclk_MHz_min": min(sc) if sc else None}
        means = [v["power_W_mean"] for v in cs["phases"].values() if v["power_W_mean"] is not None]
        swing = (max(means) - min(means)) if means else 0.0
        cs["power_swing_W"] = swing
        summ["cards"][card] = cs
        if swing > best_swing:
            best, best_swing = card, swing
    summ["active_card"] = best
    if best is not None:
        summ["phases"] = summ["cards"][best]["phases"]
        summ["power_cap_W"] = summ["cards"][best]["power_cap_W"]
        summ["cards"] = {k: {"power_swing_W": v["power_swing_W"]} for k, v in summ["cards"].items()}
    summ["command_output"] = [l.rstrip() for l in lines[-12:]]
    json.dump(summ, open(prefix + ".json", "w"), indent=1)
    print(json.dumps(summ, indent=1))
    return p.returncode


if __name__ == "__main__":
    sys.exit(main())
