"""Samples `rocm-smi` (socket power, sclk, mclk, temperature) while a command runs, and prints the samples.

    python bench/power_trace.py OUT.csv -- python bench.py --steps 200 --warmup 4 --latency-rows 0

A reading is what the driver reports for the card (an average over its own window), once every ~0.3 s; the
program under test is started as a child process (this process makes no GPU call of its own).
"""
import json
import subprocess
import sys
import time


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showuse", "--json"], capture_output=True,
                             text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "junction", "gpu use", "fclk"))}
    except Exception as e:  # noqa: BLE001 - a sampler must not end the run
        return {"error": repr(e)}


def main():
    out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    t0 = time.time()
    child = subprocess.Popen(cmd)
    rows = []
    while child.poll() is None and time.time() - t0 < 600:
        s = sample()
        s["t"] = round(time.time() - t0, 2)
        rows.append(s)
        time.sleep(0.3)
    with open(out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    sys.exit(child.returncode or 0)


if __name__ == "__main__":
    main()
