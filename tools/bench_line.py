"""One short line per bench.py run:  python tools/bench_line.py [bench.py flags...]  (runs bench.py with the side measurements off)"""
import json
import subprocess
import sys
flags = ["--no-verify", "--no-other-mode", "--no-cpu-baseline", "--no-other-dtypes", "--no-projection"]
out = subprocess.run([sys.executable, "bench.py", *flags, *sys.argv[1:]], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print(" ".join(sys.argv[1:]), "| %.0f Mpx/s  %.3f ms/step  level0 %.4f ms/launch  job frac %.3f" % (
        d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["job_roofline_frac"]), flush=True)
except Exception as e:  # noqa: BLE001
    print("bench failed:", e, out.stderr[-800:])
