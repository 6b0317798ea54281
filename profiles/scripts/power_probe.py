#!/usr/bin/env python3
"""Is the pair kernel power-limited?  Runs one layer back to back for a few seconds while a thread samples the GPU's shader clock and
socket power (sysfs hwmon, falling back to rocm-smi), for random operands and for all-zero operands, and prints the kernel time next to
the clock it ran at.

    python profiles/scripts/power_probe.py [--layer conv2] [--hilo] [--seconds 3]"""
import argparse, glob, json, os, subprocess, sys, threading, time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def sysfs_paths():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("power1_average", "power_uw"), ("power1_input", "power_uw"), ("freq1_input", "sclk_hz"), ("temp1_input", "temp_mc"), ("temp2_input", "temp_junction_mc")):
            p = os.path.join(hw, name)
            if os.path.exists(p) and key not in out:
                out[key] = p
    return out


def sample(paths):
    r = {}
    for k, p in paths.items():
        try:
            r[k] = int(open(p).read().strip())
        except Exception:
            pass
    if not r:
        try:
            j = json.loads(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = next(iter(j.values()))
            for k, v in c.items():
                if "sclk" in k.lower() and "(" in str(v):
                    r["sclk_hz"] = int(float(str(v).split("(")[1].split("M")[0]) * 1e6)
                if "power" in k.lower():
                    try:
                        r["power_uw"] = int(float(v) * 1e6)
                    except Exception:
                        pass
        except Exception as e:
            r["err"] = str(e)[:80]
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    import hesic_amd
    from compressai.layers import GDN
    from compressai.models.utils import conv
    hesic_amd.set_compute_dtype(torch.float16)
    torch.manual_seed(0)
    B, S = args.batch, args.size
    layer, g = conv(128, 128, stride=2).cuda(), GDN(128).cuda()
    paths = {}      # the hwmon nodes in this container belong to other cards of the node: rocm-smi sees the visible one
    print("sysfs:", paths)
    flops = 3 * (2.0 * B * (S // 2) ** 2 * 128 * 128 * 25 + 2.0 * B * (S // 2) ** 2 * 128 * 128)
    idle = sample(paths)
    print("idle", idle)
    for label, scale in (("random", 0.5), ("zeros", 0.0), ("random", 0.5)):
        xf = torch.randn(B, 128, S, S, device="cuda") * scale
        hi = xf.to(torch.float16)
        xh = torch.cat((hi, (xf - hi.float()).to(torch.float16)), 1).contiguous(memory_format=torch.channels_last)
        f = lambda: layer.run_hilo(xh, gdn=g)
        with torch.no_grad():
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                f()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(50):
                    f()
            samples, stop = [], threading.Event()
            def poll():
                while not stop.is_set():
                    samples.append((time.time(), sample(paths)))
                    time.sleep(0.02)
            th = threading.Thread(target=poll)
            th.start()
            t0 = time.time()
            times = []
            while time.time() - t0 < args.seconds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) / 50 * 1e3)
            stop.set(); th.join()
        half = samples[len(samples) // 2:]
        def avg(k):
            v = [x[1][k] for x in half if k in x[1]]
            return sum(v) / len(v) if v else float("nan")
        us = sorted(times)[len(times) // 2]
        print("%-7s kernel %.1f us  (first %.1f, last %.1f)  %.0f TFLOP/s executed   sclk %.0f MHz  power %.0f W  temp %.0f C  [%d samples]"
              % (label, us, times[0], times[-1], flops / us / 1e6, avg("sclk_hz") / 1e6, avg("power_uw") / 1e6, avg("temp_junction_mc") / 1e3 if "temp_junction_mc" in paths else avg("temp_mc") / 1e3, len(samples)))
        time.sleep(1.0)


if __name__ == "__main__":
    main()
